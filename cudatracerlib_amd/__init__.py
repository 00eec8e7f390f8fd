"""cudatracerlib_amd — MI355X-native wavefront path tracer behind the CudaTracerLib tracer-plugin API.

The product is ``libctl_amd.so`` (hand-written HIP for gfx950 + C++ host, C-ABI in ``include/ctl_amd.h``).
This package is a thin ctypes mirror of that C-ABI using the reference's class and method names
(``DynamicScene``, ``Image``, ``WavefrontPathTracer.Resize/InitializeScene/DoPass`` …, Kernel/Tracer.h:67-294).
There is no CPU fallback: without the built extension, importing the API raises.
"""
from .api import (  # noqa: F401
    lib, CtlError, DynamicScene, Scene, Image, Comm, WavefrontPathTracer, PathTracer, SequenceGenerator,
    ctl_material, ctl_texture, ctl_light, ctl_sensor, ctl_scene_desc, ctl_ray, ctl_hit, ctl_pixel_data,
    ctl_tracer_stats, ctl_traversal_counts, ctl_float4x4,
    diffuse, dielectric, conductor, roughconductor, device_count, intersect, intersect_count,
)
