// shade_basic.hip — shade kernel for scenes that only use the BSDFs, textures and emitters of the basic feature set (shading.h).
// CTL_BASIC_SHADE_BLOCK / CTL_BASIC_SHADE_WAVES: workgroup size and waves per SIMD the register allocation is held to (measured choices in DESIGN.md §3):
// 512-lane workgroups at 4 waves per SIMD = 128 VGPRs, two workgroups per CU — the three barriers of block_append3 then stall half a CU instead of all of it
// (1024 lanes: 1.72 ms per pass on synthetic-SM, 512: 1.60, 256: 1.61; profiles/r02r_shade_block_ab.log).
// this build: pathIterateKernel's own path rules (tracer parameter PathSemantics = Wavefront, shade_kernel.inc)
#define CTL_SHADE_WAVEFRONT_RULES 1
#define CTL_SHADE_FEATURES 0
#ifndef CTL_BASIC_SHADE_BLOCK
#define CTL_BASIC_SHADE_BLOCK 512
#endif
#define CTL_SHADE_BLOCK CTL_BASIC_SHADE_BLOCK
#ifndef CTL_BASIC_SHADE_WAVES
#define CTL_BASIC_SHADE_WAVES 4
#endif
#define CTL_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(CTL_BASIC_SHADE_WAVES, CTL_BASIC_SHADE_WAVES)))
#ifndef CTL_BASIC_SORT_WINDOW
#define CTL_BASIC_SORT_WINDOW 0   // regrouping off in this build (measured: DESIGN.md §3)
#endif
#define CTL_SHADE_SORT_WINDOW CTL_BASIC_SORT_WINDOW
#define CTL_SHADE_KERNEL k_shade_basic_wf
#define CTL_SHADE_LAUNCH launch_shade_basic_wf
#ifndef CTL_SHADE_LDS_TABLES
#define CTL_SHADE_LDS_TABLES 12   // as shade_basic.hip
#endif
#include "shade_kernel.inc"
