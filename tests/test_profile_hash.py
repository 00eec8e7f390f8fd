"""The counter-side numbers of bench.py's roofline blocks come from a COMMITTED rocprofv3 profile (profiles/roofline_traffic.json).  What ties that profile to the binary is a
hash over the kernel sources (bench.py KERNEL_BUILD / SHADE_BUILD), written into the profile by the profiled run itself (tools/profile_round.sh -> tools/summarize_profile.py).
An edit to one of those sources without a re-profile makes bench.py report "unprofiled" instead of another binary's counters — and makes this test fail, so that the
tree is not committed in that state."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    return m


def test_hash_follows_the_sources(tmp_path, monkeypatch):
    b = _bench()
    assert b.KERNEL_BUILD == "trav-" + b.source_hash(b.TRAVERSAL_SOURCES) and len(b.KERNEL_BUILD) == 17
    for n in b.TRAVERSAL_SOURCES + b.SHADE_SOURCES:
        assert os.path.exists(os.path.join(ROOT, "cudatracerlib_amd", "csrc", n)), n
    # the files whose edits change what the profiled kernels do (advisor, round 5): the BVH2 builder with its re-optimisation pass, the host code that lays the device scene out,
    # the shared structs, every *_wf build stub
    for n in ("flatten.cpp", "traverse_flat.h", "bvh_builder.cpp", "tracer.hip", "device_scene.h", "flat8.h", "knobs.h"):
        assert n in b.TRAVERSAL_SOURCES, n
    for n in ("shade_kernel.inc", "tracer.hip", "device_scene.h", "shade_basic_wf.hip", "shade_class_p_wf.hip"):
        assert n in b.SHADE_SOURCES, n
    # a copy of the sources in a scratch tree: one more statement in one file is another hash, one more COMMENT is not, another compiler flag is
    import shutil
    src = os.path.join(ROOT, "cudatracerlib_amd", "csrc"); dst = tmp_path / "cudatracerlib_amd" / "csrc"; dst.mkdir(parents=True)
    for n in set(b.TRAVERSAL_SOURCES): shutil.copy(os.path.join(src, n), dst / n)
    shutil.copy(os.path.join(ROOT, "cudatracerlib_amd", "build.py"), tmp_path / "cudatracerlib_amd" / "build.py")
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    h0 = b.source_hash(b.TRAVERSAL_SOURCES)
    assert "trav-" + h0 == b.KERNEL_BUILD
    f = dst / "bvh_builder.cpp"; text = f.read_text()
    f.write_text(text + "\n// a remark\n/* another */\n")
    assert b.source_hash(b.TRAVERSAL_SOURCES) == h0
    f.write_text(text + "\nstatic int one_more_statement = 1;\n")
    assert b.source_hash(b.TRAVERSAL_SOURCES) != h0
    f.write_text(text)
    flags = b.build_flags()
    assert "--offload-arch=gfx950" in flags and "-ffp-contract=off" in flags
    monkeypatch.setattr(b, "build_flags", lambda: flags + ["-O2"])
    assert b.source_hash(b.TRAVERSAL_SOURCES) != h0
    # bench.py itself must not load the package (libctl_amd.so and its HIP runtime) at import time: see build_flags
    import subprocess, sys
    code = ("import importlib.util, sys; s = importlib.util.spec_from_file_location('b', %r); m = importlib.util.module_from_spec(s); s.loader.exec_module(m); "
            "assert 'cudatracerlib_amd' not in sys.modules and 'torch' not in sys.modules; print(m.KERNEL_BUILD)") % os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == _bench().KERNEL_BUILD, out.stderr[-500:]


def test_comment_stripping_leaves_literals_alone():
    b = _bench()
    t = b.strip_comments('int a = 1; // c\nconst char* s = "http://x"; /* b */ int b;\n\n   \nchar c = \'"\'; // "q\n')
    assert t == 'int a = 1;\nconst char* s = "http://x";   int b;\nchar c = \'"\';'


def test_committed_profile_is_of_this_tree():
    b = _bench()
    t = json.load(open(os.path.join(ROOT, "profiles", "roofline_traffic.json")))
    key = "synthetic-sm 1920x1080 depth 8 2000 inst subdiv 4 flat"          # bench.py's default workload (BASELINE's headline stand-in)
    assert key in t["workloads"]
    e = t["workloads"][key]
    assert e["kernel_build"] == b.KERNEL_BUILD, "the traversal sources changed after profiles/%s_* was taken: re-run tools/profile_round.sh" % e.get("tag")
    assert e["shade"]["kernel_build"] == b.SHADE_BUILD, "the shade sources changed after profiles/%s_* was taken: re-run tools/profile_round.sh" % e.get("tag")
    # an entry of another build is not quoted
    assert b.calibrated_traffic(key) is not None
    assert b.calibrated_traffic("no such workload") is None
