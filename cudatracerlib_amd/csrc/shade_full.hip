// shade_full.hip — shade kernel with every BSDF, texture and emitter type built in.
#define CTL_SHADE_FEATURES 0x7F
#define CTL_SHADE_BLOCK 512
#define CTL_SHADE_KERNEL k_shade_full
#define CTL_SHADE_LAUNCH launch_shade_full
#include "shade_kernel.inc"
