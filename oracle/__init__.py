"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes access to the CPU restatement (oracle/liboracle.so) and, when built, to the reference-backed library
(oracle/_ref/libctlref.so, compiled from /root/reference's own sources by `make -C oracle ref`).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(cudatracerlib_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
f32, u32, u16, u64 = C.c_float, C.c_uint32, C.c_uint16, C.c_uint64


def build(ref=True, quiet=True):
    """compile the C++ restatement (and, when /root/reference is present, the reference-backed _ref library)"""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so", "liboracle_sm.so"], stdout=out)
    if ref and os.path.isdir("/root/reference"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=out)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(f32))


def load(shared_math=False):
    """liboracle.so: the restatement with glibc's transcendental functions — the reference's CPU path, pinned bit for bit on the reference's own code (tests/golden).
    shared_math=True: liboracle_sm.so, the same code with the product's shared fp32 functions (cudatracerlib_amd/csrc/ctl_fmath.h), which the HIP kernels run as well —
    the checker of the GPU parity tests (no libm-vs-device-library last-bit differences left to flip a discrete decision)."""
    path = os.path.join(_HERE, "liboracle_sm.so" if shared_math else "liboracle.so")
    if not os.path.exists(path):
        build(ref=False)
    o = C.CDLL(path)
    o.orc_half_to_float.restype = f32; o.orc_half_to_float.argtypes = [u16, C.c_int]
    o.orc_float_to_half.restype = u16; o.orc_float_to_half.argtypes = [f32]
    o.orc_normal_encode.restype = u16
    o.orc_normal_decode.argtypes = [u16, C.c_void_p]
    o.orc_seqgen_create.restype = C.c_void_p
    o.orc_seqgen_destroy.argtypes = [C.c_void_p]
    o.orc_seqgen_compute.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_xorwow_init.argtypes = [u64, u64, C.c_void_p]
    o.orc_render.restype = u64
    o.orc_render.argtypes = [C.c_void_p, u32, u32, u32, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, u32, u32, C.c_int]
    o.orc_intersect.argtypes = [C.c_void_p, C.c_void_p, u32, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    o.orc_set_flat_bvh.argtypes = [C.c_void_p]; o.orc_set_flat_bvh.restype = None
    o.orc_compute_partials.argtypes = [C.c_void_p] * 8
    o.orc_sensor_sample_ray_differential.argtypes = [C.c_void_p, f32, f32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_mip_eval.argtypes = [C.c_void_p, f32, f32, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_mip_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u32]; o.orc_mip_pyramid.restype = u32
    o.orc_render_counting.argtypes = [C.c_int]; o.orc_render_counting.restype = None
    o.orc_render_counts.argtypes = [C.c_void_p]; o.orc_render_counts.restype = None
    o.orc_fresnel_dielectric_ext.restype = f32; o.orc_fresnel_dielectric_ext.argtypes = [f32, f32, C.c_void_p]
    o.orc_fresnel_conductor_exact.argtypes = [f32, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_sensor_sample_ray.argtypes = [C.c_void_p, f32, f32, C.c_void_p, C.c_void_p]
    o.orc_sampler_float.restype = f32; o.orc_sampler_float.argtypes = [C.c_void_p, C.c_void_p, u32, u32]
    o.orc_sampler_float2.argtypes = [C.c_void_p, C.c_void_p, u32, u32, C.c_void_p]
    for n in ("orc_square_to_cosine_hemisphere", "orc_square_to_uniform_triangle", "orc_square_to_uniform_disk_concentric"):
        getattr(o, n).argtypes = [f32, f32, C.c_void_p]
    o.orc_microfacet_eval.argtypes = [C.c_int, f32, f32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_microfacet_sample.argtypes = [C.c_int, f32, f32, C.c_int, C.c_void_p, f32, f32, C.c_void_p]
    o.orc_bsdf_sample.argtypes = [C.c_void_p, C.c_void_p, f32, f32, C.c_void_p]
    o.orc_bsdf_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u32, C.c_void_p]
    o.orc_light_sample_direct.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p, f32, f32, C.c_void_p]
    o.orc_rough_transmittance_eval.restype = f32; o.orc_rough_transmittance_eval.argtypes = [u32, f32, f32, f32]
    o.orc_spline_eval_2d.restype = f32; o.orc_spline_eval_2d.argtypes = [f32, f32, C.c_void_p, u32, u32]
    o.orc_spline_eval_3d.restype = f32; o.orc_spline_eval_3d.argtypes = [f32, f32, f32, C.c_void_p, u32, u32, u32]
    o.orc_rough_transmittance_eval_diffuse.restype = f32; o.orc_rough_transmittance_eval_diffuse.argtypes = [u32, f32, f32]
    o.orc_set_probe_rough_transmittance.argtypes = [C.c_void_p]
    o.orc_set_probe_materials.argtypes = [C.c_void_p]
    o.orc_bsdf_eval_discrete.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u32, C.c_void_p]
    o.orc_light_pdf_direct.restype = f32
    o.orc_light_pdf_direct.argtypes = [C.c_void_p, u32, C.c_void_p, C.c_void_p, C.c_void_p, f32, C.c_void_p]
    o.orc_env_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    o.orc_texture_eval.argtypes = [C.c_void_p, C.c_void_p, f32, f32, C.c_void_p]
    o.orc_set_block_counts.argtypes = [C.c_void_p, u32]
    o.orc_set_zero_stop_image.argtypes = [C.c_void_p]; o.orc_set_zero_stop_image.restype = None
    o.orc_image_add_samples.argtypes = [C.c_void_p, u32, u32, C.c_int, C.c_void_p]; o.orc_image_add_samples.restype = None
    o.orc_sample_normal_map.argtypes = [C.c_void_p, C.c_void_p, f32, f32, C.c_void_p, C.c_void_p]
    o.orc_alpha_test.argtypes = [C.c_void_p, C.c_void_p, f32, f32]
    o.orc_triangle_data_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u32, C.c_int, C.c_void_p]
    o.orc_triangle_fill_dg.argtypes = [C.c_void_p, C.c_void_p, f32, f32, C.c_int, C.c_void_p]
    o.orc_woop_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, f32, f32, C.c_void_p]
    return o


def load_ref():
    """the reference-backed library, or None when it has not been built (no /root/reference on this box and no prebuilt file)"""
    path = os.path.join(_HERE, "_ref", "libctlref.so")
    if not os.path.exists(path):
        return None
    r = C.CDLL(path, mode=os.RTLD_LAZY)   # CUDA-runtime symbols of never-called reference code stay unresolved
    r.ref_half_to_float.restype = f32; r.ref_half_to_float.argtypes = [u16]
    r.ref_float_to_half.restype = u16; r.ref_float_to_half.argtypes = [f32]
    r.ref_normal_encode.restype = u16
    r.ref_normal_decode.argtypes = [u16, C.c_void_p]
    r.ref_fresnel_dielectric_ext.restype = f32; r.ref_fresnel_dielectric_ext.argtypes = [f32, f32, C.c_void_p]
    r.ref_fresnel_conductor_exact.argtypes = [f32, C.c_void_p, C.c_void_p, C.c_void_p]
    r.ref_power_heuristic.restype = f32; r.ref_power_heuristic.argtypes = [f32, f32]
    r.ref_sensor_sample_ray.argtypes = [C.c_void_p, f32, f32, f32, C.c_int, C.c_int, f32, f32, C.c_void_p, C.c_void_p]
    for n in ("ref_square_to_cosine_hemisphere", "ref_square_to_uniform_triangle", "ref_square_to_uniform_disk_concentric"):
        getattr(r, n).argtypes = [f32, f32, C.c_void_p]
    r.ref_microfacet_eval.argtypes = [C.c_int, f32, f32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    r.ref_microfacet_sample.argtypes = [C.c_int, f32, f32, C.c_int, C.c_void_p, f32, f32, C.c_void_p]
    r.ref_triangle_data_pack.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, u32, C.c_void_p]
    r.ref_triangle_fill_dg.argtypes = [C.c_void_p, C.c_void_p, f32, f32, C.c_void_p]
    r.ref_woop_intersect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, f32, C.c_void_p]
    r.ref_construct_bvh.argtypes = [C.c_void_p, C.c_void_p, u32, u32, C.c_void_p, C.c_void_p]
    r.ref_construct_bvh_fetch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(r, "ref_trace_two_level"):
        r.ref_trace_two_level.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, u32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if hasattr(r, "ref_wrap_coordinates"):
        r.ref_wrap_coordinates.argtypes = [f32, f32, f32, f32, C.c_int, C.c_void_p]
        r.ref_checkerboard_select.argtypes = [f32] * 6
    if hasattr(r, "ref_float3_to_rgbe"):
        r.ref_float3_to_rgbe.restype = u32; r.ref_float3_to_rgbe.argtypes = [f32, f32, f32]
        r.ref_float3_to_rgbcol.restype = u32; r.ref_float3_to_rgbcol.argtypes = [f32, f32, f32]
        r.ref_rgbe_to_float3.argtypes = [u32, C.c_void_p]; r.ref_rgbcol_to_float3.argtypes = [u32, C.c_void_p]
    if hasattr(r, "ref_filter_evaluate"):
        r.ref_filter_evaluate.restype = f32; r.ref_filter_evaluate.argtypes = [C.c_int, f32, f32, f32, f32, f32, f32]
    if hasattr(r, "ref_sensor_rays"):
        r.ref_sensor_rays.argtypes = [C.c_int, C.c_void_p, f32, f32, f32, C.c_int, C.c_int, f32, f32, f32, f32, f32, f32, f32, C.c_void_p, C.c_void_p]
        r.ref_compute_partials_origins.argtypes = [C.c_void_p] * 11
    if hasattr(r, "ref_compute_partials"):
        r.ref_compute_partials.argtypes = [C.c_void_p] * 9
        r.ref_sensor_sample_ray_differential.argtypes = [C.c_void_p, f32, f32, f32, C.c_int, C.c_int, f32, f32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    return r


class Oracle:
    """convenience wrappers over liboracle.so working on a ctypes ctl_scene_desc (host pointers)"""

    def __init__(self, shared_math=False):
        self.shared_math = shared_math
        self.lib = load(shared_math)

    def intersect(self, desc, rays, any_hit=False, count=False, threads=8, alpha_test=False, flat=None):
        """flat: a ctl_flat_bvh_desc (cudatracerlib_amd.FlatBvh(...).desc) -> traverse the product's flattened BVH instead of the two-level structure"""
        r = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        hits = np.zeros(len(r), dtype=[("dist", "f4"), ("node_idx", "i4"), ("tri_idx", "i4"), ("u", "f4"), ("v", "f4")])
        cnt = (u64 * 5)()
        self.lib.orc_set_flat_bvh(C.addressof(flat) if flat is not None else None)
        try:
            self.lib.orc_intersect(C.addressof(desc), r.ctypes.data, len(r), hits.ctypes.data, (1 if any_hit else 0) | (2 if alpha_test else 0), C.addressof(cnt) if count else None, threads)
        finally:
            self.lib.orc_set_flat_bvh(None)
        if count:
            return hits, dict(n_inner=cnt[0], n_tri=cnt[1], n_inst=cnt[2])
        return hits

    def render(self, desc, width, height, n_passes=1, tables=None, direct=True, max_path_length=8, rr_start=5, threads=8, rows=None, half_host_quirk=False, alpha_test=False, block_counts=None,
               flat=None, counts=None, partials=False, regularization=False, wavefront_rules=False, u16_barycentrics=False, omit_last_nee=False, zero_stop=None):
        """pathKernel2<DIRECT,false> over all pixels (Integrators/PathTracer.cu:182-194). tables = list of (t1, t2) per pass or None.
        alpha_test: traceRay<USE_ALPHA> when the scene has alpha maps (what the reference's single-ray path does; its wavefront
        intersectKernel has no alpha test).
        regularization: PathTraceRegularization (Integrators/PathTracer.cu:115-173), the PathTracer plugin's Regularization = true; implies partials.
        wavefront_rules: the reference wavefront kernel's own path rules (pathIterateKernel, WavefrontPathTracer.cu:51-164; ocore.h pathTraceWavefront) instead of PathTrace's;
            u16_barycentrics: hit barycentrics through the traversal result's 16-bit pair (TraceHelper.cu:722-731).
        partials: first-hit ray differentials + filtered (trilinear / EWA) texture lookups, as the megakernel PathTracer does (PathTracer.cu:60-61).
        flat: a ctl_flat_bvh_desc -> every ray walks the product's flattened BVH (same hits, other visiting order).
        counts: a dict that receives the traversal statistics of this render (path_rays, path_inner, path_tri, path_inst, occ_rays, ...).
        zero_stop: a float32 array (h, w, 7), added to: the samples the reference drops as NaN / infinite AFTER the path's throughput had become exactly zero, each as the
            radiance collected up to that vertex (what the product's kernels, which end such a path at once, count) — kernels' frame == returned frame + zero_stop.
        Returns (pixel_data (h, w, 7), rays)."""
        img = np.zeros((height, width, 7), np.float32)
        y0, y1 = (0, height) if rows is None else rows
        if tables is not None:
            t1 = np.ascontiguousarray(np.concatenate([np.asarray(t[0], np.float32).ravel() for t in tables]))
            t2 = np.ascontiguousarray(np.concatenate([np.asarray(t[1], np.float32).ravel() for t in tables]))
            assert len(tables) == n_passes
            p1, p2 = t1.ctypes.data, t2.ctypes.data
        else:
            p1 = p2 = None
        if block_counts is not None:   # samples per 64x64 block for these passes (what a block sampler decided), row-major blocks
            bc = np.ascontiguousarray(block_counts, np.uint8).ravel()
            self.lib.orc_set_block_counts(bc.ctypes.data, (width + 63) // 64)
        self.lib.orc_set_flat_bvh(C.addressof(flat) if flat is not None else None)
        if zero_stop is not None:
            assert zero_stop.dtype == np.float32 and zero_stop.shape == (height, width, 7) and zero_stop.flags["C_CONTIGUOUS"]
            self.lib.orc_set_zero_stop_image(zero_stop.ctypes.data)
        if counts is not None:
            self.lib.orc_render_counting(1)
        try:
            rays = self.lib.orc_render(C.addressof(desc), width, height, n_passes, p1, p2, 1 if direct else 0, max_path_length, rr_start,
                                       img.ctypes.data, threads, y0, y1, (1 if half_host_quirk else 0) | (2 if alpha_test else 0) | (4 if partials else 0) | (8 if regularization else 0) | (16 if wavefront_rules else 0) | (32 if u16_barycentrics else 0) | (64 if omit_last_nee else 0))
        finally:
            self.lib.orc_set_flat_bvh(None)
            self.lib.orc_set_zero_stop_image(None)
            if counts is not None:
                c8 = (u64 * 8)()
                self.lib.orc_render_counts(c8)
                self.lib.orc_render_counting(0)
                for i, k in enumerate(("path_rays", "path_inner", "path_tri", "path_inst", "occ_rays", "occ_inner", "occ_tri", "occ_inst")):
                    counts[k] = counts.get(k, 0) + int(c8[i])
            if block_counts is not None:
                self.lib.orc_set_block_counts(None, 0)
        return img, int(rays)

    def sequence_tables(self, n_passes):
        g = self.lib.orc_seqgen_create()
        out = []
        for _ in range(n_passes):
            t1 = np.zeros(4096 * 30, np.float32); t2 = np.zeros(4096 * 30 * 2, np.float32)
            self.lib.orc_seqgen_compute(g, t1.ctypes.data, t2.ctypes.data)
            out.append((t1, t2))
        self.lib.orc_seqgen_destroy(g)
        return out


def ref_trace_two_level(r, desc, rays):
    """The reference's own TracerayTemplate (both levels) + float4x4 transforms + TriIntersectorData::Intersect over the arrays of a ctl_scene_desc
    (oracle/ref_driver.cpp).  rays: (n, 8) = origin, tmin, direction, tmax; the reference's Intersect has the fixed tmin 1e-4, so the rays must carry it.
    Returns the same record array Oracle.intersect returns (tri_idx / node_idx = -1 on a miss, dist = tmax)."""
    import numpy as np
    q = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
    assert (q[:, 3] == np.float32(1e-4)).all(), "TriIntersectorData::Intersect tests t > 1e-4"
    n = len(q)
    tuv = np.zeros((n, 3), np.float32); tri = np.zeros(n, np.int32); node = np.zeros(n, np.int32)
    r.ref_trace_two_level(desc.scene_bvh_nodes, desc.scene_start_node, desc.bvh_nodes, desc.woop, desc.woop_index, desc.nodes, desc.meshes, desc.node_inv_transforms,
                          n, q.ctypes.data, tuv.ctypes.data, tri.ctypes.data, node.ctypes.data)
    hits = np.zeros(n, dtype=[("dist", "f4"), ("node_idx", "i4"), ("tri_idx", "i4"), ("u", "f4"), ("v", "f4")])
    hits["dist"] = tuv[:, 0]; hits["u"] = tuv[:, 1]; hits["v"] = tuv[:, 2]; hits["tri_idx"] = tri; hits["node_idx"] = node
    return hits
