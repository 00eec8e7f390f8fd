// shade_basic.hip — shade kernel for scenes that use only diffuse / dielectric / conductor / roughconductor BSDFs, constant and
// checkerboard textures, area and point lights (no spills at 128 VGPRs; the benchmark scene and the Cornell configs run this).
#define CTL_SHADE_FEATURES 0
#define CTL_SHADE_BLOCK 1024
#define CTL_SHADE_KERNEL k_shade_basic
#define CTL_SHADE_LAUNCH launch_shade_basic
#include "shade_kernel.inc"
