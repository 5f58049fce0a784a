// Second instantiation of the prepacked 1x16 path: 16-element codebook vectors (32 B), 32 slices of 2048 entries (64 KiB)
// x 8 row groups.  Same source as the 8-element build; see the head of gemv_packed.hip.  Entry points: aqlm_hip_g16_*,
// reached through the public aqlm_hip_* entries (in_group_size 16 / a descriptor with slices_log2 == 5).
#undef AQLM_PK_G
#undef AQLM_PK_S_LOG
#undef AQLM_PK_NG_LOG
#undef AQLM_PK_XFIRST
#define AQLM_PK_G 16
#define AQLM_PK_S_LOG 5
#define AQLM_PK_NG_LOG 3
#include "gemv_packed.hip"
