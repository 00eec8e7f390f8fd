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


def test_hash_follows_the_sources(tmp_path):
    b = _bench()
    assert b.KERNEL_BUILD == "trav-" + b.source_hash(b.TRAVERSAL_SOURCES) and len(b.KERNEL_BUILD) == 17
    for n in b.TRAVERSAL_SOURCES + b.SHADE_SOURCES:
        assert os.path.exists(os.path.join(ROOT, "cudatracerlib_amd", "csrc", n)), n
    assert "flatten.cpp" in b.TRAVERSAL_SOURCES and "traverse_flat.h" in b.TRAVERSAL_SOURCES and "shade_kernel.inc" in b.SHADE_SOURCES
    # one byte more in one file: another hash
    import hashlib
    h0 = b.source_hash(["flatten.cpp"])
    h = hashlib.sha256(); h.update(b"flatten.cpp\0"); h.update(open(os.path.join(ROOT, "cudatracerlib_amd", "csrc", "flatten.cpp"), "rb").read() + b" "); h.update(b"\1")
    assert h.hexdigest()[:12] != h0


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
