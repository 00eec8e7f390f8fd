"""Build libctl_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m cudatracerlib_amd.build          # build if stale
    python -m cudatracerlib_amd.build --force

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the tree.
"""
import hashlib, os, subprocess, sys, glob, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libctl_amd.so")
SRCS = ["kernels.hip", "shade_basic.hip", "shade_full.hip", "shade_class_a.hip", "shade_class_b.hip", "shade_class_c.hip", "shade_class_p.hip", "shade_class_g.hip", "shade_class_g_wf.hip", "shade_class_p_wf.hip", "shade_class_a_wf.hip", "shade_class_b_wf.hip", "shade_class_c_wf.hip", "shade_basic_wf.hip", "shade_full_wf.hip", "megakernel.hip", "image_pipeline.hip", "block_sampler.hip", "tracer.hip", "capi.hip", "scene_builder.cpp", "bvh_builder.cpp", "sbvh_builder.cpp", "flatten.cpp", "mitsuba_loader.cpp", "image_io.cpp", "jpeg_decode.cpp", "mesh_io.cpp", "scene_cache.cpp", "comm.cpp"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-pthread"]


KNOBS_OUT = os.path.join(HERE, "libctl_knobs.so")   # the same library built -DCTL_MEASUREMENT_KNOBS (csrc/knobs.h): tools/exp.sh, tools/*probe*, the tests of builder options


def _flag_key(defines=()):
    """what besides the sources decides the binary: the -D set and $CTL_BUILD_EXTRA_FLAGS"""
    return hashlib.sha256(("\0".join(sorted(defines)) + "\1" + os.environ.get("CTL_BUILD_EXTRA_FLAGS", "")).encode()).hexdigest()[:16]


def _object_dir(out):
    """the product's objects stay in-tree (build/, git- and gpurun-ignored); a variant's objects go under $TMPDIR — only its .so travels to the GPU box"""
    if out is None:
        return os.path.join(HERE, "build")
    return os.path.join(os.environ.get("TMPDIR") or tempfile.gettempdir(), "ctl_build_" + os.path.splitext(os.path.basename(out))[0])


def stale(target=None, defines=()):
    target = target or OUT
    if not os.path.exists(target):
        return True
    key = os.path.join(_object_dir(None if target == OUT else target), "flags.key")
    if not os.path.exists(key) or open(key).read().strip() != _flag_key(defines):
        return True   # built with another -D / flag set (or by an older build.py): an A/B run must never compare a binary with itself
    t = os.path.getmtime(target)
    deps = glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(CSRC, "experiments", "*")) + [os.path.join(HERE, "..", "include", "ctl_amd.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, out=None, defines=()):
    """out / defines: a variant build (tools/build_variant.sh) with extra -D flags into its own object directory"""
    if not force and not stale(out, defines):
        return out or OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    bdir = _object_dir(out)
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in SRCS:
        o = os.path.join(bdir, s + ".o")
        cmd = [hipcc] + FLAGS + os.environ.get("CTL_BUILD_EXTRA_FLAGS", "").split() + ["-D" + d for d in defines] + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    bad = [s for s, p in procs if p.wait() != 0]
    if bad:
        raise RuntimeError("hipcc failed for: " + ", ".join(bad))
    target = OUT if out is None else out
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", target] + objs + ["-pthread", "-lz", "-ldl"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(os.path.join(bdir, "flags.key"), "w") as f:
        f.write(_flag_key(defines) + "\n")
    return target


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    out = None; defines = []
    while args:
        a = args.pop(0)
        if a == "--out": out = os.path.abspath(args.pop(0))
        elif a.startswith("-D"): defines.append(a[2:])
    build(force="--force" in sys.argv, out=out, defines=defines, verbose=False)
    if out is None:
        build(force="--force" in sys.argv, out=KNOBS_OUT, defines=["CTL_MEASUREMENT_KNOBS"], verbose=False)
