#!/bin/bash
# Host-side sanitizer run (GPU AddressSanitizer is not available on the pool: sanitizers run on the CPU build only): libaqlm_cpu.so and the
# C oracle rebuilt with -fsanitize=address,undefined, the CPU kernel tests and the oracle's golden tests under them.  Restores the
# ordinary libraries afterwards.   bash tools/asan_cpu.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd); T=$(mktemp -d)
cp "$R/aqlm_amd/libaqlm_cpu.so" "$T/cpu.orig"; cp "$R/oracle/libaqlm_oracle.so" "$T/oracle.orig"
trap 'cp "$T/cpu.orig" "$R/aqlm_amd/libaqlm_cpu.so"; cp "$T/oracle.orig" "$R/oracle/libaqlm_oracle.so"; rm -rf "$T"' EXIT
g++ -O1 -g -std=c++17 -fPIC -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -shared -o "$R/aqlm_amd/libaqlm_cpu.so" "$R/aqlm_amd/csrc_cpu/aqlm_cpu.cpp"
gcc -O1 -g -fPIC -fopenmp -fsanitize=address,undefined -shared -o "$R/oracle/libaqlm_oracle.so" "$R/oracle/aqlm_oracle.c" -lm
cd "$R"
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) \
  python -m pytest tests/test_cpu_path.py tests/test_oracle_golden.py -x -q -p no:cacheprovider
