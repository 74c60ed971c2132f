"""
bench.py as the driver invokes it: `python bench.py --gpus N` must start N ranks itself (the reference has no
distributed entry point at all, train_cgcn.sh:5 picks one GPU).  CPU tests cover the refusals that need no GPU; the `gpu` test
runs the real self-launch with two ranks sharing the one GPU of the box over gloo (GCNPT_BENCH_ONE_DEVICE=1).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
FAST = ["--no-cpu-baseline", "--no-kernel-breakdown", "--no-pooled-only", "--no-secondary", "--no-secondary-shapes"]


def _run(args, env=None, timeout=600):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)


def test_gpus_must_agree_with_world_size():
    """Under a launcher (WORLD_SIZE set) a different --gpus is refused before anything touches a GPU."""
    r = _run(["--gpus", "2"], env={"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and b"must agree" in r.stderr, r.stderr[-500:]


def test_self_launch_refuses_more_ranks_than_gpus():
    """No GPU here: --gpus 2 must not silently run one rank (round 1 did: it printed n_gpus 1 and exited 0)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    env = {k: "" for k in ()}
    r = _run(["--gpus", "2"] + FAST, env=env)
    assert r.returncode != 0 and r.stdout.strip() == b"" and b"GPU(s) visible" in r.stderr, (r.returncode, r.stderr[-500:])


@pytest.mark.gpu
def test_self_launch_two_ranks_one_device():
    """The real entry: the parent starts 2 fresh ranks (before any GPU call of its own), they share GPU 0 and exchange the flat
    gradient bucket over gloo.  Same shard on both ranks, so the reduced bucket must be 2 x the one-rank bucket."""
    common = ["--steps", "40", "--warmup", "5", "--launch", "native"] + FAST
    one = _run(["--gpus", "1"] + common, env={"GCNPT_BENCH_SAME_SHARD": "1"})
    assert one.returncode == 0, one.stderr[-2000:]
    l1 = [ln for ln in one.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(l1) == 1
    d1 = json.loads(l1[0])
    assert d1["n_gpus"] == 1
    two = _run(["--gpus", "2", "--dist-backend", "gloo"] + common,
               env={"GCNPT_BENCH_ONE_DEVICE": "1", "GCNPT_BENCH_SAME_SHARD": "1"})
    assert two.returncode == 0, two.stderr[-3000:]
    l2 = [ln for ln in two.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(l2) == 1, two.stdout[-2000:]                              # ONE JSON line, from rank 0
    d2 = json.loads(l2[0])
    assert d2["n_gpus"] == 2 and d2["config"]["world_size"] == 2 and d2["config"]["global_batch"] == 2 * d1["config"]["global_batch"]
    assert d2["config"]["ranks_started_by"].startswith("bench.py itself")
    assert len(d2["config"]["ms_per_step_by_rank"]) == 2
    assert d2["scaling"] == "weak" and d2["value"] > 0
    g1, g2 = d1["config"]["grad_bucket_abs_sum"], d2["config"]["grad_bucket_abs_sum"]
    assert g1 > 0 and abs(g2 - 2 * g1) <= 2e-3 * g2, (g1, g2)            # float atomics reorder the sums: not bit-exact
    # the N > 1 headline is SYNCHRONOUS SGD: every step's all-reduce feeds a device-side update the next step's pack reads
    assert d2["config"]["dp_mode"].startswith("synchronous SGD") and d2["config"]["weight_abs_drift_after_timed_steps"] > 0
    assert "async_upper_bound" in d2 and "NOT synchronous" in d2["async_upper_bound"]["note"]
    assert d1["config"]["dp_mode"] == "none (1 GPU)"
