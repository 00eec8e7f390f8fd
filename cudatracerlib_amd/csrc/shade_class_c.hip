// shade_class_c.hip — model-class build of the shade kernel: the nesting models — coating, rough coating, blend — whose inner model can be any other (all models compiled in, the inner calls out of line).
// A scene that needs the full feature set AND has the traversal's key per ray (flattened BVH, dev_scene::flat_leaf_keys) is shaded by one launch per model class present in it
// (kernels.hip launch_shade), each over the slot list k_class_partition made for the class, instead of one kernel over all slots that carries every model and regroups them
// behind workgroup barriers: 256-lane workgroups, full waves of (mostly) one model, no wave that idles at a barrier while the slowest model of the workgroup finishes.
#define CTL_SHADE_FEATURES 0x7F
#define CTL_FMATH_OUTLINE   // ctl_math.h: one out-of-line copy of each transcendental function instead of one per call site (class b: 293 -> 159 KB of code against a 64-KB instruction cache)
#include "kernels.h"
#define CTL_SHADE_KEYS CTL_CLASS_C_KEYS
#define CTL_SHADE_CLASS 2
#define CTL_LIGHT_INLINE   // shading.h: no out-of-line emitter function takes the scene by reference — out of line they cost a private copy of the dev_scene argument (496 B of scratch per lane, read back with vector loads): synthetic-bathroom shade 2.52 -> 2.26 ms per pass
#define CTL_SHADE_MODELS 0xFFFFu
#ifndef CTL_CLASS_C_BLOCK
#define CTL_CLASS_C_BLOCK 256
#endif
#define CTL_SHADE_BLOCK CTL_CLASS_C_BLOCK
#ifndef CTL_CLASS_C_WAVES
#define CTL_CLASS_C_WAVES 4
#endif
#if CTL_CLASS_C_WAVES > 0
#define CTL_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(CTL_CLASS_C_WAVES, CTL_CLASS_C_WAVES)))
#endif
#if defined(CTL_SHADE_WAVEFRONT_RULES) && CTL_SHADE_WAVEFRONT_RULES   // shade_class_c_wf.hip: pathIterateKernel's own path rules (PathSemantics = Wavefront)
#define CTL_SHADE_KERNEL k_shade_class_c_wf
#define CTL_SHADE_LAUNCH launch_shade_class_c_wf
#else
#define CTL_SHADE_KERNEL k_shade_class_c
#define CTL_SHADE_LAUNCH launch_shade_class_c
#endif
#ifndef CTL_SHADE_LDS_TABLES
#define CTL_SHADE_LDS_TABLES 12   // KB of LDS for the emitter records + anim blob (shading.h scene_lights / scene_anim; + 4 KB for the normal table): synthetic-SM shade 1.447 -> 1.356 ms per pass
#endif
#include "shade_kernel.inc"
