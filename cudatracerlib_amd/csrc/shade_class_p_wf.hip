// shade_class_p_wf.hip — shade_class_p.hip with pathIterateKernel's own path rules (tracer parameter PathSemantics = Wavefront, shade_kernel.inc)
#define CTL_SHADE_WAVEFRONT_RULES 1
#include "shade_class_p.hip"
