// shade_basic.hip — shade kernel for scenes that only use the BSDFs, textures and emitters of the basic feature set (shading.h).
// CTL_BASIC_SHADE_BLOCK / CTL_BASIC_SHADE_WAVES: workgroup size and waves per SIMD the register allocation is held to (measured choices, EXPERIMENTS.md):
// 256-lane workgroups at 4 waves per SIMD = 128 VGPRs, four workgroups per CU — the three barriers of block_append3 then stall a quarter of a CU
// (round 5, with the LDS tables, shade ms per pass on synthetic-SM, workgroup / regroup window: 512 / 128 1.359, 256 / 128 1.363, 256 / 256 1.326 (and the traversal of the
// next bounce 4.92 against 4.97: its queue order), 512 / 256 1.38, 512 / 512 1.40; 5 waves per SIMD 1.70; profiles/r05t_shade_block_window.jsonl).
#define CTL_SHADE_FEATURES 0
#ifndef CTL_BASIC_SHADE_BLOCK
#define CTL_BASIC_SHADE_BLOCK 256
#endif
#define CTL_SHADE_BLOCK CTL_BASIC_SHADE_BLOCK
#ifndef CTL_BASIC_SHADE_WAVES
#define CTL_BASIC_SHADE_WAVES 4
#endif
#define CTL_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(CTL_BASIC_SHADE_WAVES, CTL_BASIC_SHADE_WAVES)))
#ifndef CTL_BASIC_SORT_WINDOW
#define CTL_BASIC_SORT_WINDOW 256   // round 4: lanes regroup by BSDF model inside windows of 128 slots (round 5: 256, with 256-lane workgroups), keyed by the byte the closest-hit traversal leaves per ray (dev_scene::hit_key_out): shade 1.52 -> 1.47 ms per
                                   // pass on synthetic-SM (256: 1.48; 512: 1.52 — a wider window packs the rough-conductor lanes better and scatters the path-state reads more; 0 = off: 1.52).  With the key
                                   // derived in the kernel (hit -> node -> triangle -> material: four dependent loads) the regrouping LOST 11 % in round 2.  The model-ordered slot lists of
                                   // the class builds (k_class_partition, 16384-slot windows, 256-lane workgroups) lose here: 2.21 ms — this kernel is bound by its path-state streams, and a list
                                   // turns them into gathers (profiles/r04_shade_experiments.log)
#endif
#define CTL_SHADE_SORT_WINDOW CTL_BASIC_SORT_WINDOW
#define CTL_SHADE_KERNEL k_shade_basic
#define CTL_SHADE_LAUNCH launch_shade_basic
#ifndef CTL_SHADE_LDS_TABLES
#define CTL_SHADE_LDS_TABLES 12   // KB of LDS for the emitter records + anim blob (shading.h scene_lights / scene_anim; + 4 KB for the normal table): synthetic-SM shade 1.447 -> 1.356 ms per pass
#endif
#include "shade_kernel.inc"
