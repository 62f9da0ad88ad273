"""bench.py as the driver calls it: `python bench.py --gpus N` with NO launcher must start its own N ranks (one per
GPU), print exactly one JSON line from rank 0, and carry the result of every transport of the per-datum reduction
inside that line (`transports`, with the RCCL communicator's own rank count).  On a 1-GPU box the N > 1 legs run in
QSMC_BENCH_SHARE_GPU=1 mode (every rank on device 0, gloo + shared memory: control flow, not a measurement)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--particles", "200000", "--steps", "12", "--warmup", "3", "--no-other-configs", "--no-cpu-baseline"]


def _run_bench(args, env_extra, timeout=900):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    return r


def _one_json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_launcherless_without_gpus_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = _run_bench(["--gpus", "2"] + SMALL, {})
    assert r.returncode != 0 and "GPU(s) visible" in (r.stderr + r.stdout)


@pytest.mark.gpu
@pytest.mark.parametrize("n_ranks", [2, 8])
def test_launcherless_shared_gpu(n_ranks):
    r = _run_bench(["--gpus", str(n_ranks)] + SMALL, {"QSMC_BENCH_SHARE_GPU": "1"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == n_ranks and line["steps"] == 12 and line["warmup"] == 3
    assert line["config"]["particles_total"] == 200000 * n_ranks
    assert line["config"]["launcher"] == "self"
    assert line["value"] > 0 and line["scaling"] == "weak"
    tr = line["transports"]
    assert tr["shm"]["per_datum_collective"] == "host shared memory" and tr["shm"]["headline"]
    assert tr["shm"]["value"] == line["value"]
    assert "skipped" in tr["rccl"]                     # one device: no RCCL communicator over it
    assert abs(line["posterior_mean"] - 0.3) < 0.2


@pytest.mark.gpu
def test_rccl_pass_inside_the_line_world1():
    """One rank through the full sharded path, the RCCL-transport pass forced: its result sits INSIDE the JSON line,
    with the rank count read back from the communicator."""
    r = _run_bench(["--gpus", "1", "--force-comm"] + SMALL, {"QSMC_BENCH_FORCE_RCCL_PASS": "1", "MASTER_PORT": "29643"})
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = _one_json_line(r.stdout)
    tr = line["transports"]
    assert tr["shm"]["headline"] and tr["shm"]["value"] == line["value"]
    rc = tr["rccl"]
    assert "error" not in rc, rc
    assert rc["ranks_in_comm"] == 1 and rc["value"] > 0 and "RCCL" in rc["per_datum_collective"]
    assert rc["resamples"] == tr["shm"]["resamples"]
    assert rc["posterior_mean"] == tr["shm"]["posterior_mean"]       # both transports sum in rank order: same bits
