#!/usr/bin/env python3
"""bench.py -- particle-updates/sec of the SMC hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE config 2 = SURVEY C2): SimplePrecessionModel, 1e7 particles PER GPU, fp64,
LiuWestResampler(a=0.98), resample_thresh 0.5, prior U[0,1], true omega = 0.3, experiment
schedule t_k = (9/8)^k (k = 0..199, wrapping with a prior reset), outcomes simulated once on the
host from a fixed seed (they do not depend on N).  A "step" is one `SMCUpdater.update(outcome_k,
t_k)` -- exactly what the reference's perf_test times (perf_testing.py:250-251) -- INCLUDING any
resample it triggers.  The cloud is resident in HBM before the timed region; device RNG (Philox).

One JSON line on rank 0 with `value` = N_total * K / wall, plus
  roofline:     the fused update kernel's achieved algorithmic HBM bytes/s (24 B/particle: read x,
                read w, write w; 16 B for the first update after a resample, whose uniform weights
                are implicit) from HIP-event kernel durations measured inside the timed region;
  cpu_baseline: the CPU oracle (NumPy restatement of the reference, oracle/np_oracle.py) timed
                here on the host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "python-qinfer_amd"))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
BYTES_PER_PARTICLE_UPDATE = 24  # 16 + 8 d, d = 1
N_SCHEDULE = 200


def schedule():
    ts = (9.0 / 8.0) ** np.arange(N_SCHEDULE, dtype=np.float64)
    rs = np.random.RandomState(0)
    pr0 = np.cos(0.3 * ts / 2) ** 2
    outcomes = (rs.random_sample(N_SCHEDULE) >= pr0).astype(np.int64)
    return ts, outcomes


def cpu_baseline(n_particles, n_data):
    """Time the oracle (kind 'port': NumPy restatement of the reference path) on this host."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import np_oracle as orc
    ts, outcomes = schedule()
    np.random.seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        smc = orc.OracleSMC(orc.precession_model(), n_particles, lambda n: np.random.random((n, 1)))
        t0 = time.perf_counter()
        for k in range(n_data):
            smc.update(int(outcomes[k]), {"t": ts[k:k + 1]})
        wall = time.perf_counter() - t0
    return {"value": n_particles * n_data / wall, "unit": "particle-updates/s", "cores": 1, "kind": "port",
            "sample": "oracle/np_oracle.py OracleSMC, SimplePrecession, N=%d, first %d data of the same "
                      "schedule (%d resamples), %.1f s wall, single-threaded NumPy" % (
                          n_particles, n_data, smc.resample_count, wall)}


def load_traffic():
    """HBM bytes per launch of the update kernel from a committed rocprofv3 --pmc pass, if any."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(path):
        try:
            return json.load(open(path)).get("update_kernel_bytes_per_launch")
        except Exception:  # noqa: BLE001
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--particles", type=float, default=1e7, help="particles PER GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-particles", type=float, default=2e6)
    ap.add_argument("--cpu-data", type=int, default=120)
    ap.add_argument("--event-stride", type=int, default=8,
                    help="put hipEvents on every N-th launch of each timed kernel kind (1 = all)")
    ap.add_argument("--force-comm", action="store_true",
                    help="run the sharded code path even with one rank (validation on a 1-GPU box)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # validation hook for a 1-GPU box: QSMC_BENCH_SHARE_GPU=1 puts every rank on device 0 and talks over gloo, so
    # that the multi-rank control flow of this file can be exercised without N GPUs (not a measurement mode)
    share_gpu = os.environ.get("QSMC_BENCH_SHARE_GPU") == "1"
    if share_gpu:       # the resampler's redraw kernel needs all its workgroups resident: split the GPU's 512 slots
        os.environ.setdefault("QSMC_REDRAW_BLOCKS", str(max(16, 512 // max(world, 2))))
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    comm = None
    if world > 1 or args.force_comm:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        from qinfer_amd.parallel import ParticleShardGroup
        comm = ParticleShardGroup()

    import qinfer_amd as qi
    from qinfer_amd.engine import get_engine
    eng = get_engine()
    n = int(args.particles)
    ts, outcomes = schedule()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n, qi.UniformDistribution([0, 1]),
                            device_rng=True, seed=0, comm=comm)
        for k in range(args.warmup):                       # untimed: first W data of a throwaway pass
            upd.update(int(outcomes[k % N_SCHEDULE]), ts[k % N_SCHEDULE:k % N_SCHEDULE + 1])
        upd.reset()
        upd._resample_count = 0
        # a launch that carries start/stop events drains the queue around itself (measured: 10.7 us per step
        # with every launch timed), so every EVENT_STRIDE-th launch of each kernel kind is timed
        eng.set_profiling(0 if os.environ.get("QSMC_BENCH_NO_EVENTS") else args.event_stride)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            k = i % N_SCHEDULE
            if i and k == 0:
                upd.reset()
            upd.update(int(outcomes[k]), ts[k:k + 1])
        barrier()
        wall = time.perf_counter() - t0
        # every update kernel of the timed region carried start/stop events (hipExtLaunchKernelGGL, on the
        # launch stream); their durations are read here, once, not per step
        all_ms, tags = eng.profile_read()
        eng.set_profiling(False)
        # tag 0: update with explicit weights (24 B/particle), 2: first update after a reset/resample, weights
        # implicit (16 B/particle), 1: the resampler's sampling kernel
        full_ms, ones_ms, sampler_ms = all_ms[tags == 0], all_ms[tags == 2], all_ms[tags == 1]

    # collective in sharded mode (the moments are all-gathered if the last step resampled): every rank calls it
    posterior_mean = float(upd.est_mean()[0])
    wall_t = torch.tensor([wall], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
    if world > 1:
        torch.distributed.all_reduce(wall_t, op=torch.distributed.ReduceOp.MAX)
    wall = float(wall_t.item())

    # RCCL prints a version banner through C stdio, which (not a tty) would be flushed at exit -- after the
    # JSON line.  Push whatever C stdio holds to stderr now so that the JSON line is the last line of stdout.
    import ctypes
    sys.stdout.flush()
    saved = os.dup(1)
    os.dup2(2, 1)
    ctypes.CDLL(None).fflush(None)
    os.dup2(saved, 1)
    os.close(saved)

    if rank == 0 and os.environ.get("QSMC_BENCH_NO_EVENTS"):
        # diagnostic only (not the bench line): the same loop without the per-kernel events
        print(json.dumps({"diagnostic": "no kernel events", "ms_per_step": wall / args.steps * 1e3}), flush=True)
        return
    if rank == 0:
        n_total = n * world
        # (under local placement a rank's shard size floats by ~1e-3 relative; n is exact at N = 1)
        full_bytes, ones_bytes = float(BYTES_PER_PARTICLE_UPDATE * n), float(16 * n)
        avg_kernel_s = float(full_ms.mean()) * 1e-3          # the dominant variant: reads x and w, writes w
        achieved = full_bytes / avg_kernel_s / 1e9
        ones_info = None
        if len(ones_ms):
            ones_s = float(ones_ms.mean()) * 1e-3
            ones_info = {"timed_launches": int(len(ones_ms)), "algorithmic_bytes_per_launch": ones_bytes,
                         "avg_kernel_us": ones_s * 1e6, "achieved": ones_bytes / ones_s / 1e9}
        line = {
            "metric": "particle-updates/sec", "value": n_total * args.steps / wall,
            "unit": "particle-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "SimplePrecessionModel SMCUpdater.update, %.0e particles/GPU, fp64, "
                                   "Liu-West a=0.98, t_k=(9/8)^k" % n,
                       "particles_per_gpu": n, "particles_total": n_total,
                       "resamples_in_timed_region": upd.resample_count, "rng": "philox4x32-10 (device)",
                       "parallelism": "particle-shard x%d" % world,
                       "per_datum_collective": (None if comm is None else
                                                ("host shared memory" if comm._host is not None else "backend all-gather")),
                       "rebalances_in_timed_region": (None if comm is None else comm.n_rebalances)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": load_traffic(),
                         "kernel": "k_update_fused<PRECESSION,VEC=2,ONES=false>", "avg_kernel_us": avg_kernel_s * 1e6,
                         "timed_launches": int(len(full_ms)), "event_stride": args.event_stride,
                         "algorithmic_bytes_per_launch": full_bytes,
                         "implicit_uniform_weight_variant": ones_info},
            "posterior_mean": posterior_mean,
        }
        if len(sampler_ms):
            # the resampler's main kernel, same clock: reads w (8 B), gathers x (8d), writes x' (8d) per particle
            samp_s = float(sampler_ms.mean()) * 1e-3
            samp_bytes = (8 + 16 * 1) * n
            line["resample_kernel"] = {"kernel": "k_bucket_sample<D=1,512>", "timed_launches": int(len(sampler_ms)),
                                       "avg_kernel_us": samp_s * 1e6, "algorithmic_bytes_per_launch": samp_bytes,
                                       "achieved": samp_bytes / samp_s / 1e9, "unit": "GB/s",
                                       "frac": samp_bytes / samp_s / 1e9 / HBM_PEAK_GBS,
                                       "bound_in_practice": "VALU issue (Philox + Box-Muller + LDS search), see DESIGN.md 3.3"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(int(args.cpu_particles), args.cpu_data)
        print(json.dumps(line), flush=True)
    if world > 1 or args.force_comm:
        comm.close()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
