// shade_full.hip — shade kernel with every BSDF, texture and emitter type built in.
// CTL_FULL_SHADE_BLOCK / CTL_FULL_SHADE_WAVES: workgroup size and waves per SIMD the register allocation is held to.  1024 lanes = 4 waves per SIMD = 128 VGPRs:
// the kernel waits on dependent loads (material -> texture -> table) for 70 % of its wave cycles, and four resident waves with 154 spilled registers
// beat two waves without spills by 20 % (synthetic-bathroom); the wider workgroup also gives the regrouping step a wider window.
#define CTL_SHADE_FEATURES 0x7F
#ifndef CTL_FULL_SHADE_BLOCK
#define CTL_FULL_SHADE_BLOCK 1024
#endif
#define CTL_SHADE_BLOCK CTL_FULL_SHADE_BLOCK
#ifdef CTL_FULL_SHADE_WAVES
#define CTL_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(CTL_FULL_SHADE_WAVES, CTL_FULL_SHADE_WAVES)))
#endif
#define CTL_SHADE_KERNEL k_shade_full
#define CTL_SHADE_LAUNCH launch_shade_full
#ifndef CTL_SHADE_LDS_TABLES
#define CTL_SHADE_LDS_TABLES 12   // KB of LDS for the emitter records + anim blob (shading.h scene_lights / scene_anim; + 4 KB for the normal table): synthetic-SM shade 1.447 -> 1.356 ms per pass
#endif
#include "shade_kernel.inc"
