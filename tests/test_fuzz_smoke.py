"""A short run of tools/fuzz_loaders.py inside the CPU suite: every seed file (one per decoder / reader / the scene loader) loads, and a few hundred mutants of
them end in a CtlError or a successful load — never in a crash of the worker process.  The long runs against the AddressSanitizer build are a tool, not a test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mutated_files_never_crash_the_front_ends(tmp_path):
    out = str(tmp_path)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_loaders.py"), "--per-seed", "12", "--procs", "4", "--out", out, "--lib", "/nonexistent", "--seed", "11"],
                       capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " 0 findings" in r.stdout, tail
    assert "28 seeds" in r.stdout, tail


def test_every_seed_file_loads(tmp_path):
    """the fuzzing is only as deep as its seeds: each one must be a file the front-end accepts"""
    sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, ROOT)
    import fuzz_loaders as F
    import cudatracerlib_amd as ctl
    from cudatracerlib_amd import api
    d = str(tmp_path)
    wrap = ('<scene version="0.5.0"><sensor type="perspective"><film type="hdrfilm"><integer name="width" value="8"/><integer name="height" value="8"/></film></sensor>'
            '<shape type="%s"><string name="filename" value="%s"/><integer name="shapeIndex" value="0"/></shape></scene>')
    for name, kind, data, aux in F.seeds(d):
        sd = os.path.join(d, name.replace(".", "_")); os.makedirs(sd, exist_ok=True)
        for an, ab in aux.items():
            p = os.path.join(sd, an); os.makedirs(os.path.dirname(p), exist_ok=True); open(p, "wb").write(ab)
        p = os.path.join(sd, name); open(p, "wb").write(data)
        if kind == "image":
            img = api.decode_image_file(p)
            assert img.ndim == 3 and img.shape[0] > 0 and img.shape[1] > 0, name
        else:
            if kind != "xml":
                x = p + ".xml"; open(x, "w").write(wrap % (kind, name)); p = x
            sc = ctl.DynamicScene(); sc.ParseMitsubaScene(p)
            assert sc.UpdateScene().n_tri_data > 0, name
