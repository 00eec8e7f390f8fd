// shade_class_a_wf.hip — shade_class_a.hip with pathIterateKernel's own path rules (tracer parameter PathSemantics = Wavefront, shade_kernel.inc)
#define CTL_SHADE_WAVEFRONT_RULES 1
#include "shade_class_a.hip"
