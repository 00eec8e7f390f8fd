"""Build libctl_amd.so (HIP kernels + C-ABI) for gfx950 with hipcc, in-tree.

    python -m cudatracerlib_amd.build          # build if stale
    python -m cudatracerlib_amd.build --force

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box with the tree.
"""
import os, subprocess, sys, glob

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libctl_amd.so")
SRCS = ["kernels.hip", "shade_basic.hip", "shade_full.hip", "megakernel.hip", "image_pipeline.hip", "block_sampler.hip", "tracer.hip", "capi.hip", "scene_builder.cpp", "bvh_builder.cpp", "sbvh_builder.cpp", "flatten.cpp", "mitsuba_loader.cpp", "image_io.cpp", "jpeg_decode.cpp", "mesh_io.cpp", "scene_cache.cpp"]
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable", "-pthread"]


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(HERE, "..", "include", "ctl_amd.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in SRCS:
        o = os.path.join(HERE, "build", s + ".o")
        cmd = [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, subprocess.Popen(cmd)))
        objs.append(o)
    bad = [s for s, p in procs if p.wait() != 0]
    if bad:
        raise RuntimeError("hipcc failed for: " + ", ".join(bad))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", OUT] + objs + ["-pthread", "-lz"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
