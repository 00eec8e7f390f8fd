// shade_basic.hip — shade kernel for scenes that only use the BSDFs, textures and emitters of the basic feature set (shading.h).
// CTL_BASIC_SHADE_BLOCK / CTL_BASIC_SHADE_WAVES: workgroup size and waves per SIMD the register allocation is held to (measured choices in DESIGN.md §3).
#define CTL_SHADE_FEATURES 0
#ifndef CTL_BASIC_SHADE_BLOCK
#define CTL_BASIC_SHADE_BLOCK 1024
#endif
#define CTL_SHADE_BLOCK CTL_BASIC_SHADE_BLOCK
#ifdef CTL_BASIC_SHADE_WAVES
#define CTL_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(CTL_BASIC_SHADE_WAVES, CTL_BASIC_SHADE_WAVES)))
#endif
#ifndef CTL_BASIC_SORT_WINDOW
#define CTL_BASIC_SORT_WINDOW 0   // regrouping off in this build (measured: DESIGN.md §3)
#endif
#define CTL_SHADE_SORT_WINDOW CTL_BASIC_SORT_WINDOW
#define CTL_SHADE_KERNEL k_shade_basic
#define CTL_SHADE_LAUNCH launch_shade_basic
#include "shade_kernel.inc"
