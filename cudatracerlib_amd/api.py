"""ctypes binding of include/ctl_amd.h with the reference's class/method names.

Every call goes through the C-ABI of libctl_amd.so; nothing here computes radiance, traverses a BVH
or falls back to a CPU path.  Errors of the C layer are raised as ``CtlError`` carrying ``ctl_last_error()``
(the text the reference would have thrown as std::runtime_error, Defines.cpp:15-29).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $CTL_AMD_LIB names another build of the same library (tools/build_variant.sh: kernel A/B runs on the GPU box); never a fallback
_LIB_PATH = os.environ.get("CTL_AMD_LIB") or os.path.join(_HERE, "libctl_amd.so")
if not os.path.exists(_LIB_PATH):
    raise ImportError(
        "cudatracerlib_amd: %s is missing — build it with `python -m cudatracerlib_amd.build` "
        "(hipcc --offload-arch=gfx950). There is no CPU fallback." % _LIB_PATH)
lib = C.CDLL(_LIB_PATH)

f32, u32, i32, u8, u64 = C.c_float, C.c_uint32, C.c_int32, C.c_uint8, C.c_uint64
MAX_NUM_LIGHTS = 16
SAMPLER_N1 = 4096 * 30


class CtlError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("ctl error %d: %s" % (code, msg))
        self.code = code


# ---------------------------------------------------------------- structs (include/ctl_amd.h)
class ctl_texture(C.Structure):
    _fields_ = [("type", u32), ("value", f32 * 3), ("value1", f32 * 3), ("uv_scale", f32 * 2), ("uv_offset", f32 * 2), ("image", u32)]


class ctl_material(C.Structure):
    _fields_ = [("bsdf_type", u32), ("combined_type", u32), ("two_sided", u32), ("node_light_index", u32),
                ("tex", ctl_texture * 4), ("f", f32 * 8), ("u", u32 * 4),
                ("map_kind", u32), ("alpha_state", u32), ("alpha_test_scalar", f32), ("alpha_test_color", f32 * 3), ("reserved_", u32 * 2),
                ("map_tex", ctl_texture), ("alpha_tex", ctl_texture)]


class ctl_light(C.Structure):
    _fields_ = [("type", u32), ("radiance", f32 * 3), ("area_dist_index", u32), ("triangles_index", u32), ("sum_area", f32), ("count", u32),
                ("orthogonal", u32), ("node_idx", u32), ("position", f32 * 3), ("direction", f32 * 3),
                ("cutoff_angle", f32), ("beam_width", f32), ("cos_cutoff_angle", f32), ("cos_beam_width", f32), ("inv_transition_width", f32),
                ("to_world", f32 * 16), ("env_image", u32), ("env_scale", f32 * 3), ("bsphere_center", f32 * 3), ("bsphere_radius", f32),
                ("cdf_rows_index", u32), ("cdf_cols_index", u32), ("row_weights_index", u32), ("normalization", f32), ("rad_texture", ctl_texture)]


class ctl_mipmap(C.Structure):
    _fields_ = [("texels", C.c_void_p), ("width", u32), ("height", u32), ("texel_type", u32), ("wrap_mode", u32), ("filter_mode", u32)]


class ctl_rough_transmittance(C.Structure):
    _fields_ = [("trans", C.c_void_p), ("diff_trans", C.c_void_p), ("eta_samples", u32), ("alpha_samples", u32), ("theta_samples", u32),
                ("eta_min", f32), ("eta_max", f32), ("alpha_min", f32), ("alpha_max", f32)]


class ctl_sensor(C.Structure):
    _fields_ = [("type", u32), ("to_world", f32 * 16), ("fov", f32), ("near_depth", f32), ("far_depth", f32), ("resolution", f32 * 2),
                ("aperture_radius", f32), ("focus_distance", f32), ("screen_scale", f32 * 2)]


class ctl_float4x4(C.Structure):
    _fields_ = [("m", f32 * 16)]


class ctl_ray(C.Structure):
    _fields_ = [("a", f32 * 4), ("b", f32 * 4)]


class ctl_hit(C.Structure):
    _fields_ = [("dist", f32), ("node_idx", i32), ("tri_idx", i32), ("u", f32), ("v", f32)]


class ctl_pixel_data(C.Structure):
    _fields_ = [("rgb", f32 * 3), ("rgb_splat", f32 * 3), ("weight_sum", f32)]


class ctl_traversal_counts(C.Structure):
    _fields_ = [("n_inner", u64), ("n_tri", u64), ("n_inst", u64), ("wave_inner_iters", u64), ("wave_tri_iters", u64)]


class ctl_tracer_stats(C.Structure):
    _fields_ = [("rays_last_pass", u64), ("rays_total", u64), ("seconds_last_pass", C.c_double), ("seconds_total", C.c_double), ("passes_done", u32),
                ("ms_intersect", C.c_double), ("ms_shade", C.c_double), ("ms_raygen", C.c_double), ("ms_intersect_any", C.c_double),
                ("intersect_rays", u64), ("intersect_launches", u64), ("shadow_rays", u64), ("shadow_launches", u64),
                ("closest_counts", ctl_traversal_counts), ("any_counts", ctl_traversal_counts), ("fused_launches", u64), ("fused_shadow_rays", u64), ("fused_closest_rays", u64), ("ms_fused", C.c_double)]


class ctl_scene_desc(C.Structure):
    _fields_ = [("tri_data", C.c_void_p), ("n_tri_data", u32), ("woop", C.c_void_p), ("n_woop", u32), ("woop_index", C.c_void_p),
                ("bvh_nodes", C.c_void_p), ("n_bvh_nodes", u32), ("meshes", C.c_void_p), ("n_meshes", u32), ("nodes", C.c_void_p), ("n_nodes", u32),
                ("materials", C.POINTER(ctl_material)), ("n_materials", u32), ("lights", C.POINTER(ctl_light)), ("n_lights_buf", u32),
                ("anim", C.c_void_p), ("n_anim_bytes", u32), ("scene_start_node", i32), ("scene_bvh_nodes", C.c_void_p), ("n_scene_bvh_nodes", u32),
                ("node_transforms", C.c_void_p), ("node_inv_transforms", C.c_void_p), ("env_map_index", u32),
                ("box_min", f32 * 3), ("box_max", f32 * 3), ("camera", ctl_sensor), ("num_lights", u32),
                ("light_indices", u32 * MAX_NUM_LIGHTS), ("light_cdf", f32 * MAX_NUM_LIGHTS), ("ray_trace_eps", f32),
                ("images", C.POINTER(ctl_mipmap)), ("n_images", u32), ("rough_transmittance", C.POINTER(ctl_rough_transmittance))]

    # numpy views of the reference-layout arrays (host memory owned by the builder)
    def view(self, name, dtype, count, width):
        ptr = getattr(self, name)
        if not ptr or not count:
            return np.zeros((0, width), dtype)
        buf = (C.c_char * (count * width * np.dtype(dtype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dtype).reshape(count, width)


assert C.sizeof(ctl_texture) == 48 and C.sizeof(ctl_material) == 384 and C.sizeof(ctl_pixel_data) == 28 and C.sizeof(ctl_hit) == 20

lib.ctl_last_error.restype = C.c_char_p
lib.ctl_version.restype = C.c_char_p
lib.ctl_image_device_ptr.restype = C.c_void_p
lib.ctl_image_device_ptr.argtypes = [C.c_void_p]
for _n in ("ctl_builder_destroy", "ctl_scene_destroy", "ctl_image_destroy", "ctl_tracer_destroy", "ctl_sequence_generator_destroy", "ctl_comm_destroy", "ctl_flat_bvh_destroy"):
    getattr(lib, _n).restype = None
    getattr(lib, _n).argtypes = [C.c_void_p]
lib.ctl_device_malloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
lib.ctl_device_free.argtypes = [C.c_void_p]
lib.ctl_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.ctl_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.ctl_memcpy_d2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
lib.ctl_intersect_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u32, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(f32)]
lib.ctl_image_resolve_rgb.argtypes = [C.c_void_p, f32, C.c_void_p]
lib.ctl_tracer_set_param_float.argtypes = [C.c_void_p, C.c_char_p, f32]
lib.ctl_builder_add_spot_light.argtypes = [C.c_void_p, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), f32, f32]
lib.ctl_builder_add_distant_light.argtypes = [C.c_void_p, C.POINTER(f32), C.POINTER(f32), f32]
lib.ctl_builder_add_image.argtypes = [C.c_void_p, C.c_void_p, u32, u32, u32, u32, u32, C.POINTER(u32)]
lib.ctl_builder_set_environment_map.argtypes = [C.c_void_p, u32, C.POINTER(f32), C.c_void_p]
lib.ctl_builder_set_camera_lookat.argtypes = [C.c_void_p, C.POINTER(f32), C.POINTER(f32), C.POINTER(f32), f32, u32, u32]


def _check(code):
    if code != 0:
        raise CtlError(code, lib.ctl_last_error().decode("utf-8", "replace"))


def device_count():
    return int(lib.ctl_device_count())


class ctl_reconstruction_filter(C.Structure):
    _fields_ = [("type", u32), ("x_width", f32), ("y_width", f32), ("p0", f32), ("p1", f32)]


class ctl_tonemap(C.Structure):
    _fields_ = [("key", f32), ("burn", f32)]


def box_filter(xw=1.0, yw=1.0):
    """BoxFilter (SceneTypes/Filter.h:21-37)"""
    return ctl_reconstruction_filter(1, xw, yw, 0.0, 0.0)


def gaussian_filter(xw=2.0, yw=2.0, alpha=-2.0):
    """GaussianFilter (SceneTypes/Filter.h:39-69); the default alpha = -2 is the reference's"""
    return ctl_reconstruction_filter(2, xw, yw, alpha, 0.0)


def mitchell_filter(B=1.0 / 3.0, Cc=1.0 / 3.0, xw=2.0, yw=2.0):
    """MitchellFilter (SceneTypes/Filter.h:71-100)"""
    return ctl_reconstruction_filter(3, xw, yw, B, Cc)


def lanczos_filter(xw=6.0, yw=6.0, tau=3.0):
    """LanczosSincFilter (SceneTypes/Filter.h:102-131)"""
    return ctl_reconstruction_filter(4, xw, yw, tau, 0.0)


def triangle_filter(xw=2.0, yw=2.0):
    """TriangleFilter (SceneTypes/Filter.h:133-150)"""
    return ctl_reconstruction_filter(5, xw, yw, 0.0, 0.0)


def tonemap(key=0.18, burn=0.0):
    """ToneMapPostProcess (Kernel/ImagePipeline/PostProcess/ToneMapPostProcess.h:8-24)"""
    return ctl_tonemap(key, burn)


def set_cache_dir(directory):
    """Directory of the compiled-geometry cache (the reference's .xmsh role); None disables.  Default: $CTL_CACHE_DIR."""
    _check(lib.ctl_set_cache_dir(None if directory is None else str(directory).encode()))


FLAT_Q4, FLAT_F4, FLAT_F2, FLAT_Q8 = 0, 1, 2, 3          # CTL_FLAT_* node formats of the flattened BVH
FLAT_FORMATS = {"q4": FLAT_Q4, "f4": FLAT_F4, "f2": FLAT_F2, "q8": FLAT_Q8}
DEFAULT_FLAT_FORMAT = "q4"                                 # what Scene(desc, flatten=True) builds when no format is named (csrc/flatten.cpp default_flat_format)


def flatten_probe(desc, format=FLAT_Q4):
    """Host half of Scene(desc, flatten=True): dict(nodes, leaves, depth, hash) of the flattened BVH (built or loaded from the cache)."""
    out = (u64 * 4)()
    _check(lib.ctl_flatten_probe(C.byref(desc), u32(format), out))
    return dict(nodes=out[0], leaves=out[1], depth=out[2], hash=out[3])


class FlatBvhDesc(C.Structure):
    _fields_ = [("format", u32), ("max_depth", u32), ("nodes", C.c_void_p), ("n_nodes", u64), ("node_bytes", u32),
                ("leaves", C.c_void_p), ("n_leaves", u64), ("child_links", C.c_void_p), ("compact", u32), ("root_slab", u32), ("n_slab_nodes", u64)]


class FlatBvh:
    """The flattened BVH as host arrays (ctl_flat_bvh_build): what Scene(desc, flatten=True) uploads.  .desc is a ctl_flat_bvh_desc."""

    def __init__(self, desc, format=FLAT_Q4):
        self._h = C.c_void_p()
        self._keepalive = desc
        _check(lib.ctl_flat_bvh_build(C.byref(desc), u32(format), C.byref(self._h)))
        self.desc = FlatBvhDesc()
        _check(lib.ctl_flat_bvh_arrays(self._h, C.byref(self.desc)))

    def nodes(self):
        n = self.desc.n_nodes * self.desc.node_bytes // 4
        return np.ctypeslib.as_array(C.cast(self.desc.nodes, C.POINTER(C.c_uint32)), shape=(n,)).reshape(self.desc.n_nodes, -1)

    def child_links(self):
        """the explicit links: Q4 (n_nodes, 4) int32, Q8 (n_nodes, 8) in slot order"""
        w = 8 if self.desc.format == FLAT_Q8 else 4
        return np.ctypeslib.as_array(C.cast(self.desc.child_links, C.POINTER(C.c_int32)), shape=(self.desc.n_nodes * w,)).reshape(-1, w)

    def leaves(self):
        return np.ctypeslib.as_array(C.cast(self.desc.leaves, C.POINTER(C.c_uint32)), shape=(self.desc.n_leaves * 32,)).reshape(-1, 32)

    @staticmethod
    def implied_links(N):
        """Q4 nodes (n, 16) uint32 -> (n, 4) int32: the links a traversal step derives from the first 48 B (csrc/flatten.h: link = base + nibble);
        an inner link carries the child's slab flag in bit 0.  Slots without a child decode to anything."""
        w0, w1, leafm = N[:, 10].astype(np.uint32), N[:, 11].astype(np.uint32), N[:, 3] >> 28
        ib4, nlb15 = w0 & np.uint32(0x03fffffc), (w1 >> 6) | np.uint32(0xfc000000)
        t = [None, (w0 >> 26) & 15, (w1 >> 2) & 15, (w0 >> 30) | ((w1 & 3) << 2)]
        out = np.empty((len(N), 4), np.uint32)
        out[:, 0] = np.where((leafm & 1) == 1, nlb15 + np.uint32(15), w0 & np.uint32(0x03ffffff))
        for k in (1, 2, 3):
            out[:, k] = np.where(((leafm >> k) & 1) == 1, nlb15, ib4) + t[k].astype(np.uint32)
        return out.view(np.int32)

    @staticmethod
    def clear_slab_flags(N):
        """in place: no inner link hands a slab flag down any more"""
        inner = ((N[:, 3] >> 24) & 15) & ~(N[:, 3] >> 28)      # existing children without a leaf bit; bit 0 of a slot's nibble is the flag only for an inner child (a leaf child's nibble is 15 - entries before it)
        for k, (word, bit) in enumerate(((10, 0), (10, 26), (11, 2), (10, 30))):
            N[:, word] &= ~(((inner >> k) & 1).astype(np.uint32) << np.uint32(bit))
        return N

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ctl_flat_bvh_destroy(self._h)
            self._h = None


def _fp(a):
    return a.ctypes.data_as(C.POINTER(f32))


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


# ---------------------------------------------------------------- material helpers (Mitsuba plugin defaults, ObjectParser.h:754-966)
E = dict(Null=0x1, DiffuseReflection=0x2, DiffuseTransmission=0x4, GlossyReflection=0x8, GlossyTransmission=0x10, DeltaReflection=0x20, DeltaTransmission=0x40)
TEXEL_RGBE, TEXEL_RGBCOL = 0, 1
WRAP_REPEAT, WRAP_CLAMP, WRAP_MIRROR, WRAP_BLACK = 0, 1, 2, 3
FILTER_POINT, FILTER_BILINEAR, FILTER_ANISOTROPIC, FILTER_TRILINEAR = 0, 1, 2, 3
TEX_CONSTANT, BSDF_DIFFUSE, BSDF_DIELECTRIC, BSDF_CONDUCTOR, BSDF_ROUGHCONDUCTOR = 2, 1, 3, 6, 7


def _const_tex(rgb):
    t = ctl_texture()
    t.type = TEX_CONSTANT
    rgb = (rgb, rgb, rgb) if np.isscalar(rgb) else rgb
    t.value[:] = [float(x) for x in rgb]
    t.uv_scale[:] = [1.0, 1.0]
    return t


def image_texture(image, scale=1.0, uv_scale=(1.0, 1.0), uv_offset=(0.0, 0.0)):
    """ImageTexture(mapping, file, scale) (SceneTypes/Texture.h:159-183) over a registered image."""
    t = _const_tex(scale)
    t.type = 4
    t.uv_scale[:] = [float(x) for x in uv_scale]; t.uv_offset[:] = [float(x) for x in uv_offset]
    t.image = image
    return t


def checker_texture(color0, color1, uv_scale=(1.0, 1.0), uv_offset=(0.0, 0.0)):
    """CheckerboardTexture (SceneTypes/Texture.h:127-157)."""
    t = _const_tex(color0)
    t.type = 3
    c1 = (color1, color1, color1) if np.isscalar(color1) else color1
    t.value1[:] = [float(x) for x in c1]
    t.uv_scale[:] = [float(x) for x in uv_scale]; t.uv_offset[:] = [float(x) for x in uv_offset]
    return t


def _as_tex(v):
    return v if isinstance(v, ctl_texture) else _const_tex(v)


def float3_to_rgbe(rgb):
    """SpectrumConverter::Float3ToRGBE (Math/Spectrum.h:534-555) over an (h, w, 3) float array -> (h, w) uint32 texels."""
    c = np.asarray(rgb, np.float32)
    mx = c.max(axis=-1)
    m, e = np.frexp(mx.astype(np.float64))
    with np.errstate(divide="ignore", invalid="ignore"):
        k = (m.astype(np.float32) * np.float32(256.0) / mx).astype(np.float32)
    out = np.zeros(c.shape[:-1], np.uint32)
    ok = mx >= 1e-32
    q = np.zeros(c.shape, np.uint32)
    q[ok] = (c[ok] * k[ok][:, None]).astype(np.uint8)   # (unsigned char)(c * max_)
    out[ok] = q[ok][:, 0] | (q[ok][:, 1] << 8) | (q[ok][:, 2] << 16) | (((e[ok] + 128).astype(np.uint32) & 0xff) << 24)
    return out


def float3_to_rgbcol(rgb):
    """SpectrumConverter::Float3ToCOLORREF (Math/Spectrum.h:521-526)."""
    c = (np.clip(np.asarray(rgb, np.float32), 0.0, 1.0) * np.float32(255.0)).astype(np.uint8).astype(np.uint32)
    return c[..., 0] | (c[..., 1] << 8) | (c[..., 2] << 16) | np.uint32(255 << 24)


def _material(bsdf_type, combined, two_sided=False):
    m = ctl_material()
    m.bsdf_type, m.combined_type, m.two_sided, m.node_light_index = bsdf_type, combined, 1 if two_sided else 0, 0xFFFFFFFF
    for i in range(4):
        m.tex[i] = _const_tex(0.0)
    return m


def diffuse(reflectance=(0.5, 0.5, 0.5), two_sided=False):
    """diffuse(reflectance) — BSDF_Simple.h:6-24."""
    m = _material(BSDF_DIFFUSE, E["DiffuseReflection"], two_sided)
    m.tex[0] = _as_tex(reflectance)
    return m


def dielectric(int_ior=1.5046, ext_ior=1.000277, specular_transmittance=1.0, specular_reflectance=1.0):
    """dielectric(eta = intIOR / extIOR) — BSDF_Simple.h:62-94, defaults bk7 / air (Utils.h:280-307)."""
    m = _material(BSDF_DIELECTRIC, E["DeltaReflection"] | E["DeltaTransmission"])
    m.tex[0], m.tex[1] = _const_tex(specular_transmittance), _const_tex(specular_reflectance)
    m.f[0] = np.float32(np.float32(int_ior) / np.float32(ext_ior))
    m.f[1] = 0.0
    return m


def conductor(eta=(0.0, 0.0, 0.0), k=(1.0, 1.0, 1.0), specular_reflectance=1.0, two_sided=False):
    """conductor(eta, k) — BSDF_Simple.h:165-193."""
    m = _material(BSDF_CONDUCTOR, E["DeltaReflection"], two_sided)
    m.tex[0] = _const_tex(specular_reflectance)
    m.f[0:3] = [float(x) for x in eta]
    m.f[3:6] = [float(x) for x in k]
    return m


def roughconductor(alpha=0.1, eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), distribution=1, sample_visible=True, specular_reflectance=1.0, alpha_v=None, two_sided=False):
    """roughconductor(type, eta, k, alphaU, alphaV) — BSDF_Simple.h:195-232 (distribution: 0 Beckmann, 1 GGX)."""
    m = _material(BSDF_ROUGHCONDUCTOR, E["GlossyReflection"], two_sided)
    m.tex[0], m.tex[1], m.tex[2] = _const_tex(specular_reflectance), _const_tex(alpha), _const_tex(alpha if alpha_v is None else alpha_v)
    m.f[0:3] = [float(x) for x in eta]
    m.f[3:6] = [float(x) for x in k]
    m.u[0], m.u[1] = distribution, 1 if sample_visible else 0
    return m


def thindielectric(int_ior=1.5046, ext_ior=1.000277, specular_transmittance=1.0, specular_reflectance=1.0):
    """thindielectric(eta) — BSDF_Simple.h:96-125."""
    m = _material(4, E["DeltaReflection"] | E["DeltaTransmission"])   # the reference's constructor (BSDF_Simple.h:102-118); sample() still reports ENull for the transmitted lobe
    m.tex[0], m.tex[1] = _const_tex(specular_transmittance), _const_tex(specular_reflectance)
    m.f[0] = np.float32(np.float32(int_ior) / np.float32(ext_ior))
    return m


def roughdielectric(alpha=0.1, int_ior=1.5046, ext_ior=1.000277, distribution=1, sample_visible=True, specular_transmittance=1.0, specular_reflectance=1.0, alpha_v=None):
    """roughdielectric(type, eta, alphaU, alphaV) — BSDF_Simple.h:127-163 (distribution: 0 Beckmann, 1 GGX)."""
    m = _material(5, E["GlossyReflection"] | E["GlossyTransmission"])
    m.tex[0], m.tex[1] = _const_tex(specular_transmittance), _const_tex(specular_reflectance)
    m.tex[2], m.tex[3] = _const_tex(alpha), _const_tex(alpha if alpha_v is None else alpha_v)
    eta = np.float32(np.float32(int_ior) / np.float32(ext_ior))
    m.f[0], m.f[1] = eta, np.float32(1.0) / eta
    m.u[0], m.u[1] = distribution, 1 if sample_visible else 0
    return m


def fresnel_diffuse_reflectance(eta):
    """FresnelHelper::fresnelDiffuseReflectance(eta, false) (Math/FresnelHelper.cu:57-60): the library's restatement of the reference's adaptive Gauss-Lobatto
    quadrature (csrc/material_factory.h), bit-equal to the reference's value"""
    lib.ctl_fresnel_diffuse_reflectance.restype = C.c_float; lib.ctl_fresnel_diffuse_reflectance.argtypes = [C.c_float]
    return float(lib.ctl_fresnel_diffuse_reflectance(float(np.float32(eta))))


def material_update(m):
    """BSDF::Update(): the derived fields (fdrInt / fdrExt, invEta2, sampling weights) from the primary ones, by the library (ctl_material_update).  A weight that comes
    from the average of an IMAGE texture is final only after DynamicScene.UpdateScene (ctl_builder_finalize), when the bitmap is there — as in the reference, whose
    UpdateMaterialsPhase2 runs Update() after the textures are loaded (Engine/DynamicScene.cpp:74-89)."""
    _check(lib.ctl_material_update(C.byref(m)))
    return m


def plastic(diffuse_reflectance=(0.5, 0.5, 0.5), int_ior=1.49, ext_ior=1.000277, specular_reflectance=1.0, nonlinear=False):
    """plastic(eta, diffuse, specular) — BSDF_Simple.h:234-270 incl. Update(): fdrInt/fdrExt, invEta2, specularSamplingWeight."""
    m = _material(8, E["DeltaReflection"] | E["DiffuseReflection"])
    m.tex[0], m.tex[1] = _as_tex(diffuse_reflectance), _as_tex(specular_reflectance)
    m.f[2] = float(np.float32(np.float32(int_ior) / np.float32(ext_ior)))
    m.u[0] = 1 if nonlinear else 0
    return material_update(m)


def phong(diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(0.2, 0.2, 0.2), exponent=30.0):
    """phong(diffuse, specular, exponent) — BSDF_Simple.h:313-340; specularSamplingWeight = sAvg / (dAvg + sAvg)."""
    m = _material(10, E["GlossyReflection"] | E["DiffuseReflection"])
    m.tex[0], m.tex[1], m.tex[2] = _as_tex(diffuse_reflectance), _as_tex(specular_reflectance), _as_tex(exponent)
    return material_update(m)


def roughdiffuse(reflectance=(0.5, 0.5, 0.5), alpha=0.2, use_fast_approx=False):
    """roughdiffuse(reflectance, alpha) — BSDF_Simple.h:26-60 (Oren-Nayar)."""
    m = _material(2, E["DiffuseReflection"])
    m.tex[0], m.tex[1] = _as_tex(reflectance), _as_tex(alpha)
    m.u[0] = 1 if use_fast_approx else 0
    return m


def ward(diffuse_reflectance=(0.5, 0.5, 0.5), specular_reflectance=(0.2, 0.2, 0.2), alpha_u=0.1, alpha_v=0.1, variant=2):
    """ward(variant, diffuse, specular, alphaU, alphaV) — BSDF_Simple.h:342-381; variant 0 Ward, 1 Ward-Duer, 2 balanced."""
    m = _material(11, E["GlossyReflection"] | E["DiffuseReflection"])
    m.tex[0], m.tex[1], m.tex[2], m.tex[3] = _as_tex(diffuse_reflectance), _as_tex(specular_reflectance), _as_tex(alpha_u), _as_tex(alpha_v)
    m.u[0] = variant
    return material_update(m)


def roughplastic(diffuse_reflectance=(0.5, 0.5, 0.5), alpha=0.1, int_ior=1.49, ext_ior=1.000277, distribution=0, specular_reflectance=1.0, nonlinear=False, sample_visible=True):
    """roughplastic(type, eta, alpha, diffuse, specular) — BSDF_Simple.h:272-312; needs the rough-transmittance table of slot `distribution`."""
    m = _material(9, E["GlossyReflection"] | E["DiffuseReflection"])
    m.tex[0], m.tex[1], m.tex[2] = _as_tex(diffuse_reflectance), _as_tex(specular_reflectance), _as_tex(alpha)
    m.f[0] = float(np.float32(np.float32(int_ior) / np.float32(ext_ior)))
    m.u[0], m.u[1], m.u[2] = 1 if nonlinear else 0, 0 if distribution == 2 else (1 if sample_visible else 0), distribution   # getSampleVisible(type, true)
    return material_update(m)


def coating(nested_index, nested, int_ior=1.5046, ext_ior=1.000277, thickness=1.0, sigma_a=0.0, specular_reflectance=1.0):
    """coating(nested, eta, thickness, sigmaA, specular) — BSDF_Complex.h:9-75.  `nested_index` = DynamicScene.add_material(nested)."""
    m = _material(13, E["DeltaReflection"] | nested.combined_type)
    m.tex[0], m.tex[1] = _as_tex(sigma_a), _as_tex(specular_reflectance)
    eta = float(np.float32(np.float32(int_ior) / np.float32(ext_ior)))
    m.f[0], m.f[2] = eta, thickness
    m.u[2] = nested_index
    return material_update(m)


def roughcoating(nested_index, nested, alpha=0.1, int_ior=1.5046, ext_ior=1.000277, thickness=1.0, sigma_a=0.0, distribution=0, specular_reflectance=1.0):
    """roughcoating(type, nested, eta, thickness, sigmaA, alpha, specular) — BSDF_Complex.h:77-147; needs the rough-transmittance table."""
    m = _material(14, E["GlossyReflection"] | nested.combined_type)
    m.tex[0], m.tex[1], m.tex[2] = _as_tex(sigma_a), _as_tex(specular_reflectance), _as_tex(alpha)
    eta = float(np.float32(np.float32(int_ior) / np.float32(ext_ior)))
    m.f[0], m.f[2] = eta, thickness
    m.u[0], m.u[1], m.u[2] = distribution, 0 if distribution == 2 else 1, nested_index
    return material_update(m)


MAP_NONE, MAP_NORMAL, MAP_HEIGHT = 0, 1, 2
ALPHA_DISABLED, ALPHA_MAP_LUMINANCE, ALPHA_MAP_ALPHA, ALPHA_MAP_COLOR = 0, 1, 2, 3
ALPHA_REFLECTANCE_LUMINANCE, ALPHA_REFLECTANCE_ALPHA, ALPHA_REFLECTANCE_COLOR = 5, 6, 7


def set_normal_map(material, texture):
    """Material::SetNormalMap (Engine/Material.h:93-99); excludes a height map."""
    if material.map_kind == MAP_HEIGHT:
        raise CtlError(-1, "Cannot set both height and normal map!")
    material.map_kind = MAP_NORMAL; material.map_tex = _as_tex(texture)
    return material


def set_height_map(material, texture):
    """Material::SetHeightMap (Engine/Material.h:100-106); only image textures perturb the frame (Material.cu:109)."""
    if material.map_kind == MAP_NORMAL:
        raise CtlError(-1, "Cannot set both height and normal map!")
    material.map_kind = MAP_HEIGHT; material.map_tex = _as_tex(texture)
    return material


def set_alpha_map(material, texture, state=ALPHA_MAP_LUMINANCE, test_scalar=1.0, test_color=(0.0, 0.0, 0.0)):
    """Material::SetAlphaMap + AlphaBlendData::test_val_* (Engine/Material.h:24-36,107-111)."""
    material.alpha_state = state; material.alpha_tex = _as_tex(texture)
    material.alpha_test_scalar = float(test_scalar); material.alpha_test_color[:] = [float(x) for x in test_color]
    return material


def blend(index0, nested0, index1, nested1, weight=0.5):
    """blend(nested1, nested2, weight) — BSDF_Complex.h:149-181 (blendbsdf / mixturebsdf of the loader)."""
    m = _material(15, nested0.combined_type | nested1.combined_type)
    m.tex[0] = _as_tex(weight)
    m.u[2], m.u[3] = index0, index1
    return m


# ---------------------------------------------------------------- DynamicScene (Engine/DynamicScene.h:70-187, loader-facing subset)
def decode_image_file(path):
    """ctl_decode_image_file: (H, W, 3) float32 for HDR / PFM / EXR files, (H, W, 4) uint8 otherwise; rows top-down"""
    w, h, fl = u32(0), u32(0), i32(0)
    _check(lib.ctl_decode_image_file(path.encode(), C.byref(w), C.byref(h), C.byref(fl), None, None))
    if fl.value:
        out = np.zeros((h.value, w.value, 3), np.float32)
        _check(lib.ctl_decode_image_file(path.encode(), C.byref(w), C.byref(h), C.byref(fl), out.ctypes.data_as(C.POINTER(C.c_float)), None))
    else:
        out = np.zeros((h.value, w.value, 4), np.uint8)
        _check(lib.ctl_decode_image_file(path.encode(), C.byref(w), C.byref(h), C.byref(fl), None, out.ctypes.data_as(C.POINTER(C.c_uint8))))
    return out


class DynamicScene:
    def __init__(self):
        self._h = C.c_void_p()
        _check(lib.ctl_builder_create(C.byref(self._h)))
        self.desc = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ctl_builder_destroy(self._h)
            self._h = None

    def add_mesh(self, positions, indices=None, normals=None, uvs=None, tri_material=None, materials=None):
        """Mesh::CompileMesh (Engine/Mesh.cpp:199-290) -> mesh index."""
        P = _f32(positions, (-1, 3))
        I = None if indices is None else np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1, 3)
        n_tri = len(P) // 3 if I is None else len(I)
        N = None if normals is None else _f32(normals, (-1, 3))
        UV = None if uvs is None else _f32(uvs, (-1, 2))
        TM = None if tri_material is None else np.ascontiguousarray(tri_material, dtype=np.uint8)
        mats = materials if materials is not None else [diffuse()]
        arr = (ctl_material * len(mats))(*mats)
        out = u32()
        _check(lib.ctl_builder_add_mesh(self._h, _fp(P), u32(len(P)), None if I is None else I.ctypes.data_as(C.c_void_p), u32(n_tri),
                                        None if N is None else _fp(N), None if UV is None else _fp(UV),
                                        None if TM is None else TM.ctypes.data_as(C.c_void_p), arr, u32(len(mats)), C.byref(out)))
        return out.value

    def CreateNode(self, mesh_index, to_world=None):
        """DynamicScene::CreateNode + SetNodeTransform (DynamicScene.cpp:269-346)."""
        out = u32()
        m = None
        if to_world is not None:
            m = ctl_float4x4()
            m.m[:] = [float(x) for x in np.asarray(to_world, dtype=np.float32).reshape(16)]
        _check(lib.ctl_builder_add_node(self._h, u32(mesh_index), None if m is None else C.byref(m), C.byref(out)))
        return out.value

    def CreateLight(self, node, local_material, radiance, rad_texture=None, orthogonal=False):
        """DynamicScene::CreateLight(node, materialName, L) (DynamicScene.cpp:689-711); materials are addressed by local index.
        rad_texture / orthogonal: DiffuseLight::m_rad_texture (a checker or image ctl_texture) and m_bOrthogonal (Light.h:100-101)."""
        L = (f32 * 3)(*[float(x) for x in radiance])
        if rad_texture is None and not orthogonal:
            _check(lib.ctl_builder_add_area_light(self._h, u32(node), u32(local_material), L))
        else:
            _check(lib.ctl_builder_add_area_light_ex(self._h, u32(node), u32(local_material), L, C.byref(rad_texture) if rad_texture is not None else None, i32(1 if orthogonal else 0)))

    def CreatePointLight(self, position, intensity):
        _check(lib.ctl_builder_add_point_light(self._h, (f32 * 3)(*map(float, position)), (f32 * 3)(*map(float, intensity))))

    def setCamera(self, pos, target, up, fov_degrees, width, height):
        _check(lib.ctl_builder_set_camera_lookat(self._h, (f32 * 3)(*map(float, pos)), (f32 * 3)(*map(float, target)), (f32 * 3)(*map(float, up)),
                                                 f32(fov_degrees), u32(width), u32(height)))

    def CreateSpotLight(self, position, target, intensity, cutoff_angle=20.0, beam_width=None):
        """SpotLight(p, t, L, cutoffAngle, beamWidth) in degrees; Mitsuba default beamWidth = 3/4 cutoffAngle (ObjectParser.h spot)."""
        bw = cutoff_angle * 0.75 if beam_width is None else beam_width
        _check(lib.ctl_builder_add_spot_light(self._h, (f32 * 3)(*position), (f32 * 3)(*target), (f32 * 3)(*intensity), f32(cutoff_angle), f32(bw)))

    def CreateDistantLight(self, direction, irradiance, scene_radius=1.0):
        _check(lib.ctl_builder_add_distant_light(self._h, (f32 * 3)(*direction), (f32 * 3)(*irradiance), f32(scene_radius)))

    def add_image(self, texels, texel_type=TEXEL_RGBCOL, wrap=WRAP_REPEAT, filter=FILTER_BILINEAR):
        """Register level 0 of a KernelMIPMap: texels (h, w) uint32, byte 0 = r ... byte 3 = e / alpha.  Returns the image index."""
        t = np.ascontiguousarray(texels, dtype=np.uint32)
        idx = u32()
        _check(lib.ctl_builder_add_image(self._h, t.ctypes.data_as(C.c_void_p), u32(t.shape[1]), u32(t.shape[0]), u32(texel_type), u32(wrap), u32(filter), C.byref(idx)))
        return idx.value

    def set_bvh_mode(self, mode):
        """mesh BVH builder for the following add_mesh calls: "auto" (default), "sbvh" (the reference's SplitBVHBuilder restated) or "binned"."""
        _check(lib.ctl_builder_set_bvh_mode(self._h, u32({"auto": 0, "sbvh": 1, "binned": 2}[mode])))

    def add_material(self, material):
        """register the nested BSDF of a coating / roughcoating / blend; returns its absolute material index"""
        idx = u32()
        _check(lib.ctl_builder_add_material(self._h, C.byref(material), C.byref(idx)))
        return idx.value

    def setRoughTransmittance(self, slot, trans, diff_trans, eta_range, alpha_range):
        """RoughTransmittanceManager slot (0 beckmann.dat, 1 phong.dat, 2 ggx.dat — indexed by the distribution TYPE at run time,
        RoughTransmittance.cu:124-157).  trans: (2*eta, alpha, theta) float32, diff_trans: (2*eta, alpha)."""
        t = np.ascontiguousarray(trans, np.float32); dt = np.ascontiguousarray(diff_trans, np.float32)
        r = ctl_rough_transmittance()
        r.trans, r.diff_trans = t.ctypes.data, dt.ctypes.data
        r.eta_samples, r.alpha_samples, r.theta_samples = t.shape[0] // 2, t.shape[1], t.shape[2]
        r.eta_min, r.eta_max, r.alpha_min, r.alpha_max = eta_range[0], eta_range[1], alpha_range[0], alpha_range[1]
        _check(lib.ctl_builder_set_rough_transmittance(self._h, u32(slot), C.byref(r)))

    def loadRoughTransmittance(self, slot, path):
        _check(lib.ctl_builder_load_rough_transmittance(self._h, u32(slot), path.encode()))

    def setEnvironementMap(self, image, scale=(1.0, 1.0, 1.0), to_world=None):
        """DynamicScene::setEnvironementMap (DynamicScene.cpp:846-859) for an already decoded lat-long image."""
        m = None
        if to_world is not None:
            m = ctl_float4x4(); m.m[:] = [float(x) for x in np.asarray(to_world, np.float32).reshape(16)]
        _check(lib.ctl_builder_set_environment_map(self._h, u32(image), (f32 * 3)(*scale), C.byref(m) if m is not None else None))

    def setSensor(self, sensor):
        """the camera as a ctl_sensor struct (ctl_builder_set_camera), e.g. another scene's desc.camera"""
        _check(lib.ctl_builder_set_camera(self._h, C.byref(sensor)))

    def ParseMitsubaScene(self, path, width=-1, height=-1):
        """ParseMitsubaScene (Engine/SceneLoader/Mitsuba/MitsubaLoader.h:13)."""
        w, h = i32(width), i32(height)
        _check(lib.ctl_parse_mitsuba_scene(self._h, path.encode(), C.byref(w), C.byref(h)))
        return w.value, h.value

    def UpdateScene(self):
        """DynamicScene::UpdateScene + getKernelSceneData(false) (DynamicScene.cpp:480-589): host-side KernelDynamicScene."""
        self.desc = ctl_scene_desc()
        _check(lib.ctl_builder_finalize(self._h, C.byref(self.desc)))
        return self.desc

    def getKernelSceneData(self):
        return self.desc if self.desc is not None else self.UpdateScene()


class Scene:
    """The scene resident in HBM (UpdateKernel, Kernel/TraceHelper.cu:182-217)."""

    def __init__(self, desc, flatten=False, flat_format=None, reduced_rough_transmittance=False):
        """flat_format: None = the library default (Q4, or $CTL_FLAT_FORMAT), else FLAT_Q4 / FLAT_F4 / FLAT_F2 or one of the strings q4 / f4 / f2.
        reduced_rough_transmittance: CTL_SCENE_REDUCED_ROUGH_TRANSMITTANCE (faster rough plastic, equal to the reference's lookup up to rounding only)"""
        self._h = C.c_void_p()
        self._keepalive = desc
        flags = (1 if flatten else 0) | (2 if reduced_rough_transmittance else 0)
        if flat_format is not None:
            flags |= (FLAT_FORMATS.get(flat_format, flat_format) + 1) << 8
        _check(lib.ctl_scene_create_ex(C.byref(desc), u32(flags), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ctl_scene_destroy(self._h)
            self._h = None


def _rays_struct(rays):
    r = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
    return r, r.ctypes.data_as(C.c_void_p)


def intersect(scene, rays, any_hit=False):
    """__internal__IntersectBuffers (Kernel/TraceHelper.cu:736-746). rays: (n, 8) = ox oy oz tmin dx dy dz tmax.
    Returns a structured array with dist, node_idx, tri_idx, u, v."""
    r, rp = _rays_struct(rays)
    hits = np.zeros(len(r), dtype=[("dist", "f4"), ("node_idx", "i4"), ("tri_idx", "i4"), ("u", "f4"), ("v", "f4")])
    _check(lib.ctl_intersect(scene._h, rp, u32(len(r)), hits.ctypes.data_as(C.c_void_p), 1 if any_hit else 0))
    return hits


def trace_single_ray(scene, origin, direction, tmin=0.0, tmax=3.402823466e+38):
    """TracerBase::TraceSingleRay (Kernel/Tracer.cu:74-78) -> the record of the closest hit (dist, node_idx, tri_idx, u, v)"""
    rays = np.zeros((1, 8), np.float32); rays[0, :3] = origin; rays[0, 3] = tmin; rays[0, 4:7] = direction; rays[0, 7] = tmax
    r, rp = _rays_struct(rays)
    hit = np.zeros(1, dtype=[("dist", "f4"), ("node_idx", "i4"), ("tri_idx", "i4"), ("u", "f4"), ("v", "f4")])
    _check(lib.ctl_trace_single_ray(scene._h, rp, hit.ctypes.data_as(C.c_void_p)))
    return hit[0]


def intersect_count(scene, rays, any_hit=False):
    r, rp = _rays_struct(rays)
    c = ctl_traversal_counts()
    _check(lib.ctl_intersect_count(scene._h, rp, u32(len(r)), 1 if any_hit else 0, C.byref(c)))
    return dict(n_inner=c.n_inner, n_tri=c.n_tri, n_inst=c.n_inst, wave_inner_iters=c.wave_inner_iters, wave_tri_iters=c.wave_tri_iters)


class Comm:
    """The framebuffer reduce of a multi-GPU render (ctl_comm_*, comm.cpp): one rank per process and GPU, RCCL over xGMI.
    Comm.unique_id() on one rank -> the 128 bytes to every rank -> Comm(id, rank, world) on every rank (collective) -> reduce(image, root)."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _check(lib.ctl_comm_get_unique_id(buf))
        return bytes(buf)

    def __init__(self, unique_id, rank, world, timeout_ms=0):
        """timeout_ms: give ncclCommInitRank up after this long (0: $CTL_COMM_TIMEOUT_MS, else 120 s) — a rank that never arrives raises instead of hanging the job"""
        self._h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(lib.ctl_comm_create_timeout(buf, C.c_int32(rank), C.c_int32(world), C.c_int32(timeout_ms), C.byref(self._h)))

    def reduce(self, image, root=0):
        """sum of all ranks' PixelData frames into `root`'s image, in place, one ncclReduce; returns when it is complete.  ONE call per frame: a second one on an
        image that already holds a reduced frame is refused (use reduce_to for a per-pass gather)"""
        _check(lib.ctl_image_reduce(image._h, self._h, C.c_int32(root)))

    def reduce_to(self, src, dst, root=0):
        """out of place (the per-pass gather of a progressive display): `dst` on the root = sum over the ranks of `src`; every rank's src stays its own cumulative frame.
        dst may be None on the other ranks"""
        _check(lib.ctl_image_reduce_to(src._h, dst._h if dst is not None else None, self._h, C.c_int32(root)))

    def gather(self, image, root=0):
        """north_star's exchange: every rank's own tiles + halo (ceil(tiles / world) x 65 x 65 x 28 B) to `root`'s image, in place, one ncclGather; ONE call per render"""
        _check(lib.ctl_image_gather(image._h, self._h, C.c_int32(root)))

    def gather_to(self, src, dst, root=0):
        """out of place (repeatable: the per-pass progressive gather): `dst` on the root = every rank's own tiles of its `src`; dst may be None on the other ranks"""
        _check(lib.ctl_image_gather_to(src._h, dst._h if dst is not None else None, self._h, C.c_int32(root)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ctl_comm_destroy(self._h)
            self._h = None


class Image:
    """Engine/Image.h:31-91 — PixelData accumulator in HBM."""

    def __init__(self, width, height):
        self._h = C.c_void_p()
        self.width, self.height = width, height
        _check(lib.ctl_image_create(u32(width), u32(height), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ctl_image_destroy(self._h)
            self._h = None

    def packedTileBytes(self, world):
        """bytes of one rank's packed tiles (ctl_image_packed_tile_bytes)"""
        n = C.c_uint64()
        _check(lib.ctl_image_packed_tile_bytes(u32(self.width), u32(self.height), u32(world), C.byref(n)))
        return n.value

    def packTiles(self, rank, world):
        """the tiles t % world == rank of this image with their one-pixel halo as (slots, 65 * 65, 7) float32 — what ctl_image_gather sends (ctl_image_pack_tiles)"""
        out = np.empty(self.packedTileBytes(world) // 4, np.float32)
        _check(lib.ctl_image_pack_tiles(self._h, u32(rank), u32(world), out.ctypes.data_as(C.c_void_p)))
        return out.reshape(-1, 65 * 65, 7)

    def unpackTiles(self, world, packed_all_ranks):
        """write the packed tiles of ALL ranks (rank-major) into this image: tiles copied, then halos added (ctl_image_unpack_tiles)"""
        a = np.ascontiguousarray(packed_all_ranks, np.float32)
        if a.nbytes != self.packedTileBytes(world) * world:
            raise ValueError("unpackTiles: %d bytes, expected %d" % (a.nbytes, self.packedTileBytes(world) * world))
        _check(lib.ctl_image_unpack_tiles(self._h, u32(world), a.ctypes.data_as(C.c_void_p)))

    def Clear(self):
        _check(lib.ctl_image_clear(self._h))

    def getPixelData(self):
        """(h, w, 7) float32: rgb[3], rgbSplat[3], weightSum."""
        a = np.zeros((self.height, self.width, 7), np.float32)
        _check(lib.ctl_image_read_pixels(self._h, a.ctypes.data_as(C.c_void_p)))
        return a

    def addSamples(self, samples):
        """Image::AddSample (Engine/Image.cu:22-44) for samples (n, 5) = sx, sy, r, g, b"""
        a = np.ascontiguousarray(samples, dtype=np.float32).reshape(-1, 5)
        _check(lib.ctl_image_add_samples(self._h, u32(len(a)), a.ctypes.data_as(C.c_void_p)))

    def setPixelData(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(self.height, self.width, 7)
        _check(lib.ctl_image_write_pixels(self._h, a.ctypes.data_as(C.c_void_p)))

    def device_ptr(self):
        return lib.ctl_image_device_ptr(self._h)

    def applyImagePipeline(self, splat_scale=0.0, filter=None, process=None):
        """applyImagePipeline(tracer, img, filter, process) (Kernel/ImagePipeline/ImagePipeline.cu:54-84): (h, w, 4) uint8 display image.
        filter = ctl_reconstruction_filter (see box_filter ... triangle_filter) or None; process = ctl_tonemap (see tonemap) or None."""
        a = np.zeros((self.height, self.width), np.uint32)
        if filter is None and process is None:
            _check(lib.ctl_image_apply_pipeline(self._h, f32(splat_scale), a.ctypes.data_as(C.c_void_p)))
        else:
            _check(lib.ctl_image_apply_pipeline_ex(self._h, f32(splat_scale), None if filter is None else C.byref(filter),
                                                   None if process is None else C.byref(process), a.ctypes.data_as(C.c_void_p)))
        return a.view(np.uint8).reshape(self.height, self.width, 4)

    def WriteDisplayImage(self, path, splat_scale=0.0):
        """Image::WriteDisplayImage: .png (display image), .hdr / .pfm (linear)."""
        _check(lib.ctl_image_write_file(self._h, f32(splat_scale), path.encode()))

    def getRGB(self, splat_scale=0.0):
        """copySamplesToOutput (Kernel/ImagePipeline/ImagePipeline.cu:14-30) up to linear RGB."""
        a = np.zeros((self.height, self.width, 3), np.float32)
        _check(lib.ctl_image_resolve_rgb(self._h, f32(splat_scale), a.ctypes.data_as(C.c_void_p)))
        return a

class _Parameters:
    def __init__(self, tracer):
        self._t = tracer

    def setValue(self, key, value):
        """bool, int (also the index of an enum value), float, or the name of an enum value (TracerParameterCollection::setValue, Kernel/TracerSettings.h:277)"""
        if isinstance(value, bool):
            _check(lib.ctl_tracer_set_param_bool(self._t._h, key.encode(), 1 if value else 0))
        elif isinstance(value, str):
            _check(lib.ctl_tracer_set_param_enum(self._t._h, key.encode(), value.encode()))
        elif isinstance(value, float):
            _check(lib.ctl_tracer_set_param_float(self._t._h, key.encode(), f32(value)))
        else:
            _check(lib.ctl_tracer_set_param_int(self._t._h, key.encode(), int(value)))

    def getFloat(self, key):
        v = f32()
        _check(lib.ctl_tracer_get_param_float(self._t._h, key.encode(), C.byref(v)))
        return v.value

    def getValue(self, key):
        v = C.c_int()
        _check(lib.ctl_tracer_get_param_int(self._t._h, key.encode(), C.byref(v)))
        return v.value


class WavefrontPathTracer:
    """Integrators/PseudoRealtime/WavefrontPathTracer.h:24-67 behind Tracer<true> (Kernel/Tracer.h:193-294)."""

    PLUGIN = b"WavefrontPathTracer"

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib.ctl_tracer_create(self.PLUGIN, C.byref(self._h)))
        self._scene = None

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ctl_tracer_destroy(self._h)
            self._h = None
        self._free_depth()

    def _free_depth(self):
        d = getattr(self, "_depth", None)
        if d is not None:
            lib.ctl_device_free(d[0]); self._depth = None

    def getParameters(self):
        return _Parameters(self)

    def Resize(self, w, h):
        _check(lib.ctl_tracer_resize(self._h, u32(w), u32(h)))

    def InitializeScene(self, scene):
        self._scene = scene
        _check(lib.ctl_tracer_initialize_scene(self._h, scene._h))

    def reservePasses(self, n):
        """size the ray queues for a DoPasses(n) to come (they would otherwise grow inside that call)"""
        _check(lib.ctl_tracer_reserve_passes(self._h, u32(n)))

    def setBlockWeight(self, block_x, block_y, weight):
        """IUserPreferenceSampler::setWeight of the tracer's block sampler (parameter BlockSamplerType); after Resize"""
        _check(lib.ctl_tracer_set_block_weight(self._h, u32(block_x), u32(block_y), f32(weight)))

    def getBlockCounts(self, width, height):
        """samples per 64x64 block of the last rendered pass: (blocks_y, blocks_x) uint8"""
        by, bx = (height + 63) // 64, (width + 63) // 64
        a = np.zeros((by, bx), np.uint8)
        _check(lib.ctl_tracer_get_block_counts(self._h, a.ctypes.data_as(C.c_void_p), u32(bx * by)))
        return a

    def setTileShard(self, rank, world):
        _check(lib.ctl_tracer_set_tile_shard(self._h, u32(rank), u32(world)))

    def setSamplerTables(self, t1, t2):
        t1, t2 = _f32(t1), _f32(t2)
        assert t1.size == SAMPLER_N1 and t2.size == 2 * SAMPLER_N1
        _check(lib.ctl_tracer_set_sampler_tables(self._h, _fp(t1), _fp(t2)))

    def DoPass(self, image, new_trace=False):
        _check(lib.ctl_tracer_do_pass(self._h, image._h, 1 if new_trace else 0))

    def DoPasses(self, image, n, new_trace=False):
        _check(lib.ctl_tracer_do_passes(self._h, image._h, 1 if new_trace else 0, u32(n)))

    def Debug(self, image, x, y):
        """TracerBase::Debug(Image*, Vec2i) (Kernel/Tracer.h:119-123): draws the next set of sampling tables from the tracer's generator, follows one path for the pixel
        (PathTracer plugin) and returns its radiance as 3 floats"""
        out = (f32 * 3)()
        _check(lib.ctl_tracer_debug_pixel(self._h, image._h, u32(x), u32(y), out))
        return np.array(out[:], np.float32)

    def setDepthBuffer(self, width, height):
        """IDepthTracer::setDepthBuffer: allocates width*height floats on the device and hands them to the tracer; read them back with getDepthBuffer()"""
        lib.ctl_tracer_set_depth_buffer.argtypes = [C.c_void_p, C.c_void_p, u32, u32]
        if getattr(self, "_depth", None) is not None:      # a buffer set before: the tracer lets go of it first, then it is freed
            lib.ctl_tracer_set_depth_buffer(self._h, None, u32(0), u32(0)); self._free_depth()
        p = C.c_void_p()
        _check(lib.ctl_device_malloc(C.c_size_t(4 * width * height), C.byref(p)))
        try:
            _check(lib.ctl_tracer_set_depth_buffer(self._h, p, u32(width), u32(height)))
        except Exception:
            lib.ctl_device_free(p)                          # the tracer refused (the PathTracer plugin is no IDepthTracer): nothing keeps the allocation
            raise
        self._depth = (p, width, height)

    def getDepthBuffer(self):
        p, w, h = self._depth
        out = np.zeros((h, w), np.float32)
        _check(lib.ctl_device_synchronize())
        _check(lib.ctl_memcpy_d2h(out.ctypes.data_as(C.c_void_p), p, C.c_size_t(out.nbytes)))
        return out

    def setCounting(self, on):
        """count N_inner / N_tri / N_inst in the intersect kernels (measurement mode)"""
        _check(lib.ctl_tracer_set_counting(self._h, 1 if on else 0))

    def stats(self):
        s = ctl_tracer_stats()
        _check(lib.ctl_tracer_get_stats(self._h, C.byref(s)))
        return s

    def getRaysInLastPass(self):
        return self.stats().rays_last_pass

    def getLastTimeSpentRenderingSec(self):
        return self.stats().seconds_last_pass

    def getNumPassesDone(self):
        return self.stats().passes_done


class PathTracer(WavefrontPathTracer):
    """Integrators/PathTracer.h:7-31 — the megakernel integrator (one kernel per pass) behind the same plugin API; needs a
    flattened scene.  For A/B against the wavefront tracer."""
    PLUGIN = b"PathTracer"


class SequenceGenerator:
    """SamplingSequenceGeneratorHost<IndependantSamplingSequenceGenerator> (Kernel/Sampler.h:36-85)."""

    def __init__(self):
        self._h = C.c_void_p()
        _check(lib.ctl_sequence_generator_create(C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            lib.ctl_sequence_generator_destroy(self._h)
            self._h = None

    def compute(self):
        t1 = np.zeros(SAMPLER_N1, np.float32)
        t2 = np.zeros(2 * SAMPLER_N1, np.float32)
        _check(lib.ctl_sequence_generator_compute(self._h, _fp(t1), _fp(t2)))
        return t1, t2

    def compute_many(self, n, threads=8):
        """tables of the next n passes, (n, 30*4096) and (n, 2*30*4096), generated in parallel through XORWOW skip-ahead"""
        t1 = np.zeros((n, SAMPLER_N1), np.float32)
        t2 = np.zeros((n, 2 * SAMPLER_N1), np.float32)
        _check(lib.ctl_sequence_generator_compute_many(self._h, u32(n), _fp(t1), _fp(t2), u32(threads)))
        return t1, t2

    def compute_many_device(self, n):
        """the same tables written in HBM by the tracers' k_sequence_fill and copied back (needs a HIP device)"""
        t1 = np.zeros((n, SAMPLER_N1), np.float32)
        t2 = np.zeros((n, 2 * SAMPLER_N1), np.float32)
        _check(lib.ctl_sequence_generator_compute_many_device(self._h, u32(n), _fp(t1), _fp(t2)))
        return t1, t2
