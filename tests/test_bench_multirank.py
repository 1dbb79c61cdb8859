"""`bench.py --gpus N` for N > 1 (the path the driver's 8-GPU node runs): rank spawning, the world-size barrier, the MAX
all-reduce of the timed region and rank-0-only output.  Pattern: habitat-lab-dialog/test/test_ddppo_reduce.py:26-126 (spawned
workers on localhost).

CPU: `spawn_ranks` propagates a failing rank's exit code.  GPU: two ranks on the ONE visible GPU over gloo
(AVLEN_DIST_BACKEND=gloo: RCCL needs one device per rank), started as a fresh child process."""
import json
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_spawn_ranks_propagates_a_failing_rank(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    script = tmp_path / "fake_rank.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        assert os.environ["WORLD_SIZE"] == "3" and os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        assert os.environ["LOCAL_RANK"] == os.environ["RANK"]
        open(os.path.join(os.path.dirname(__file__), "rank%s" % os.environ["RANK"]), "w").write(" ".join(sys.argv[1:]))
        sys.exit(7 if os.environ["RANK"] == "1" and "--fail" in sys.argv else 0)
    """))
    monkeypatch.setattr(bench, "__file__", str(script))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3", "--fail"])
    assert bench.spawn_ranks(3) == 7
    assert sorted(p.name for p in tmp_path.glob("rank*")) == ["rank0", "rank1", "rank2"]
    assert (tmp_path / "rank2").read_text() == "--gpus 3 --fail"          # every rank gets the parent's command line
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    assert bench.spawn_ranks(3) == 0


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_over_gloo():
    env = dict(os.environ, AVLEN_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--envs", "4", "--rollout", "6", "--steps", "1",
           "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-extras"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                               # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["num_envs_per_gpu"] == 4 and out["config"]["rollout_steps"] == 6
    assert out["value"] > 0 and out["ms_per_step"] > 0
    # whole-job throughput: the env-steps of BOTH ranks over the slowest rank's time
    assert abs(out["value"] - 2 * 4 * 6 * 1 / (out["ms_per_step"] * 1e-3)) <= 0.02 * out["value"]
