#!/bin/bash
# usage: tools/isa.sh <file.hip> <mangled-kernel-substring>   -> /tmp/asm/k.s (one kernel, asm markers stripped)
set -e
mkdir -p /tmp/asm
cd /root/repo/aqlm_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=fast -S --cuda-device-only -o /tmp/asm/full.s "$1" 2>&1 | grep -v warning || true
awk -v pat="$2" 'index($0, pat) && /^_Z.*:/ {p=1} p {print} p && /s_endpgm/ {exit}' /tmp/asm/full.s | grep -v "ASMSTART\|ASMEND" > /tmp/asm/k.s
wc -l /tmp/asm/k.s
grep -A12 "\.name:.*$2" /tmp/asm/full.s | grep "vgpr_count\|sgpr_count\|group_segment\|private_segment" | head -4
