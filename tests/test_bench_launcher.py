"""`python bench.py --gpus N` launches its N ranks itself (one process per device, the reference's main.cpp:160-172 flow) — checked here without
a device through --launch-only: the spawn, the gloo rendezvous on 127.0.0.1, the communicator-id broadcast and the scalar reductions of a real run,
and ONE JSON line on stdout that says n_gpus = N.  The rendering N-rank flow on one GPU is tests/test_gpu_render.py::test_bench_self_launch_two_ranks_share_one_gpu."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=180):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_launches_that_many_ranks():
    p = _run(["--gpus", "2", "--launch-only"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                       # rank 0's line only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_joined"] == 2 and out["max_over_ranks"] == 2.0 and out["id_broadcast_ok"] and out["self_launched"]


def test_three_ranks():
    out = json.loads(_run(["--gpus", "3", "--launch-only"]).stdout)
    assert out["n_gpus"] == 3 and out["ranks_joined"] == 3


def test_eight_ranks():
    """the world size of BASELINE config 4 (8 x MI355X): eight processes rendezvous, get the communicator id and reduce their scalars"""
    out = json.loads(_run(["--gpus", "8", "--launch-only"], timeout=400).stdout)
    assert out["n_gpus"] == 8 and out["ranks_joined"] == 8 and out["max_over_ranks"] == 8.0 and out["id_broadcast_ok"]


def test_one_gpu_runs_in_process():
    out = json.loads(_run(["--gpus", "1", "--launch-only"]).stdout)
    assert out["n_gpus"] == 1 and not out["self_launched"]      # N = 1 is today's single-process code path, no launcher in between


def test_world_size_must_match_gpus():
    p = _run(["--gpus", "4", "--launch-only"], {"WORLD_SIZE": "2", "RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "1"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_more_ranks_than_devices_is_refused():
    # this container has no HIP device: a rendering run with --gpus 2 must stop in the launcher, before any rank is started
    import cudatracerlib_amd as ctl
    if ctl.device_count() >= 2:
        pytest.skip("box has two devices")
    p = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert p.returncode != 0 and "HIP device" in p.stderr and p.stdout.strip() == ""


def test_watchdog_ends_a_job_with_a_stuck_rank_and_names_it():
    """a rank wedged before a rendezvous / inside a collective: the launcher's ONE deadline (counted from the spawn) ends every rank and says which were still running"""
    import time
    t = time.time()
    p = _run(["--gpus", "3", "--launch-only", "--launch-timeout", "20"], {"CTL_BENCH_TEST_RANK_FAULT": "hang:1"}, timeout=120)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "no result within --launch-timeout 20" in p.stderr and "still running" in p.stderr and "1" in p.stderr.split("still running")[1][:60]
    assert time.time() - t < 90


def test_watchdog_ends_the_others_when_a_rank_dies():
    """a rank that exits early: the others would sit in the next barrier until gloo's own time-out; the launcher ends them and reports the exit code"""
    p = _run(["--gpus", "2", "--launch-only", "--launch-timeout", "120"], {"CTL_BENCH_TEST_RANK_FAULT": "die:1"}, timeout=120)
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "rank(s) [1] failed" in p.stderr and "7" in p.stderr
