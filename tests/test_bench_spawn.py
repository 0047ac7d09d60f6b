"""`python bench.py --gpus 2` exactly as the driver may call it WITHOUT a launcher: bench.py must start the ranks itself
(fresh child processes, before anything touches the GPU), rendezvous over 127.0.0.1, gather, and print one JSON line
from rank 0.  No GPU here, so the device solve is stubbed by the CPU build of the solver header (--stub host_twin, test
infrastructure) and the collective goes over gloo; RCCL itself can only run on a multi-GPU node."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0]), p.stderr


def test_bench_spawns_its_own_ranks_weak_scaling():
    r, err = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "96", "--stub", "host_twin", "--backend", "gloo", "--inflight", "2"])
    assert "spawned 2 ranks itself" in err and "no GPU call" in err
    assert r["n_gpus"] == 2 and r["steps"] == 3 and r["scaling"] == "weak"
    assert r["config"]["global_batch"] == 192 and r["config"]["batch_per_gpu"] == 96
    assert r["config"]["gather_checked"] is True and "gloo" in r["config"]["collective_mode"]
    assert "gather to rank 0" in r["config"]["collective_mode"] and r["config"]["batches_per_collective"] == 4     # the defaults
    assert r["config"]["gather_bytes_sent_per_rank_per_batch"] == 8 * 96 * (9 + 1 + 20)
    assert r["converged_fraction"] == 1.0 and r["value"] > 0 and "stub" in r and r["metric"].startswith("STUB")
    r, _ = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "96", "--stub", "host_twin", "--backend", "gloo", "--gather", "all",
                 "--gather-results-only", "--gather-group", "1"])
    assert "all_gather" in r["config"]["collective_mode"] and "results only" in r["config"]["collective_mode"] and r["config"]["gather_checked"] is True
    assert r["config"]["gather_bytes_sent_per_rank_per_batch"] == 8 * 96 * 10


def test_bench_strong_scaling_and_f32_flags():
    """configs[3]-style strong scaling (global batch split over the ranks) and the fp32 packed gather (two int rows)."""
    r, _ = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--scaling", "strong", "--global-batch", "128", "--N", "25", "--dt", "0.05",
                 "--config", "config-stable.json", "--stub", "host_twin", "--backend", "gloo"])
    assert r["scaling"] == "strong" and r["config"]["batch_per_gpu"] == 64 and r["config"]["global_batch"] == 128 and r["config"]["N"] == 25
    r, _ = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64", "--weights-sweep", "--precision", "f32", "--no-traj",
                 "--stub", "host_twin", "--backend", "gloo"])
    assert r["dtype"] == "f32" and r["config"]["gather_checked"] is True and r["converged_fraction"] == 1.0
    assert r["roofline"]["algorithmic_bytes_per_launch"] == 136 * 64


def test_bench_parent_refuses_nothing_but_stays_gpu_free():
    """The spawning parent asserts that neither torch nor the HIP runtime is loaded in it (bench.py:spawn_ranks)."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[:src.index("def cpu_baseline_legs")]
    assert "import torch" not in head and "libamdhip64" in head
