"""world_size-2 `gloo` tests (CPU) of the particle-sharding protocol in qinfer_amd/parallel.py:
rank-order-deterministic reductions, the shared-seed count matrix, the all-to-all of ancestor rows
and the exactness of the two-level multinomial (shard, then within-shard) against the oracle."""
import os
import socket
import sys
import traceback

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name, tmpdir):
    try:
        for p in (os.path.join(ROOT, "python-qinfer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
            sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from qinfer_amd.parallel import ParticleShardGroup
        comm = ParticleShardGroup(seed=1234)
        globals()[fn_name](comm, rank, world, tmpdir)
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # noqa: BLE001
        traceback.print_exc()
        raise


def _run(fn_name, tmp_path, world=2):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), fn_name, str(tmp_path)), nprocs=world, join=True)


# ---------------------------------------------------------------------------------------------
def _global_cloud(world, n_local, d, seed=0):
    rs = np.random.RandomState(seed)
    x = rs.randn(world * n_local, d) * 0.3 + 1.0
    w = rs.random_sample(world * n_local) ** 3
    w[: n_local // 2] *= 5.0                        # rank 0 is heavier: exercises the count matrix
    w /= w.sum()
    return x, w


def _check_reductions(comm, rank, world, tmpdir):
    import torch
    import np_oracle as orc
    n_local, d = 1500, 3
    x, w = _global_cloud(world, n_local, d)
    sl = slice(rank * n_local, (rank + 1) * n_local)
    xl, wl = x[sl], w[sl]
    # update-stats reduction
    L = np.cos(xl[:, 0]) ** 2
    wp = wl * L
    got = comm.combine_update_stats(torch.tensor([wp.sum(), (wp ** 2).sum(), wp.min(), 0.0], dtype=torch.float64))
    full = w * np.cos(x[:, 0]) ** 2
    np.testing.assert_allclose(got[0], full.sum(), rtol=1e-14)
    np.testing.assert_allclose(got[1], (full ** 2).sum(), rtol=1e-14)
    assert got[2] == full.min() and got[3] == 0.0
    # moments reduction == oracle moments of the concatenated cloud
    s0 = wl.sum()
    s1 = wl @ xl
    s2 = np.einsum('i,im,in->mn', wl, xl, xl)
    g0, g1, g2 = comm.allreduce_moments(None, s0, s1, s2)
    np.testing.assert_allclose(g0, 1.0, rtol=1e-14)
    np.testing.assert_allclose(g1, orc.particle_mean(w, x), rtol=1e-13)
    cov = g2 - np.outer(g1, g1)
    np.testing.assert_allclose(cov, orc.particle_cov(w, x, warn=False), rtol=0, atol=1e-14)
    # bitwise identical on every rank (rank-ordered summation)
    rows = comm.gather_rows(torch.from_numpy(np.concatenate([[g0], g1, g2.ravel()])))
    assert np.array_equal(rows[0], rows[1])
    np.testing.assert_allclose(comm.allreduce_scalar(None, float(rank + 1)), sum(range(1, world + 1)))
    t = comm.allreduce_tensor(torch.full((2, 3), float(rank + 1), dtype=torch.float64))
    assert float(t[0, 0]) == sum(range(1, world + 1))


def _check_resample_protocol(comm, rank, world, tmpdir):
    import torch
    n_local, d = 4000, 2
    x, w = _global_cloud(world, n_local, d, seed=3)
    sl = slice(rank * n_local, (rank + 1) * n_local)
    xl, wl = x[sl], w[sl]
    W = comm.gather_rows(torch.tensor([wl.sum()], dtype=torch.float64))[:, 0]
    counts = comm.plan_counts(W, n_local, epoch=7)
    # identical plan on both ranks; rows sum to n_local
    allc = comm.gather_rows(torch.from_numpy(counts.astype(np.float64).ravel()))
    assert np.array_equal(allc[0], allc[1])
    assert counts.shape == (world, world) and np.all(counts.sum(axis=1) == n_local)
    # a different epoch gives a different plan, the same epoch the same plan
    assert np.array_equal(counts, comm.plan_counts(W, n_local, epoch=7))
    assert not np.array_equal(counts, comm.plan_counts(W, n_local, epoch=8))
    # local draws with the ORACLE's search (numpy), tagged with their global index
    rs = np.random.RandomState(100 + rank)
    cdf = np.cumsum(wl / wl.sum())
    n_draw = int(counts[:, rank].sum())
    js = np.minimum(np.searchsorted(cdf, rs.random_sample(n_draw), side='right'), n_local - 1)
    rows = np.concatenate([xl[js], (js + rank * n_local)[:, None].astype(np.float64)], axis=1)
    recv = comm.exchange_rows(torch.from_numpy(rows), counts).numpy()
    assert recv.shape == (n_local, d + 1)
    gidx = recv[:, -1].astype(np.int64)
    np.testing.assert_array_equal(recv[:, :d], x[gidx])            # rows arrived intact
    src = gidx // n_local
    for h in range(world):                                         # ordered by source, right counts
        assert np.sum(src == h) == counts[rank, h]
    assert np.all(np.diff(src) >= 0)
    np.save(os.path.join(tmpdir, "gidx_%d.npy" % rank), gidx)
    comm.dist.barrier()
    if rank == 0:
        # exactness: pooled ancestor frequencies follow the GLOBAL weights (chi-square on 20 bins)
        g = np.concatenate([np.load(os.path.join(tmpdir, "gidx_%d.npy" % r)) for r in range(world)])
        order = np.argsort(-w)
        bins = np.array_split(order, 20)
        obs = np.array([np.isin(g, b).sum() for b in bins], dtype=float)
        exp = np.array([w[b].sum() for b in bins]) * g.size
        chi2 = ((obs - exp) ** 2 / exp).sum()
        assert chi2 < 60, chi2                                     # 19 dof: P(chi2 > 60) ~ 3e-6
        # shard-level split follows W
        frac0 = np.mean(g < n_local)
        assert abs(frac0 - W[0] / W.sum()) < 5 * np.sqrt(0.25 / g.size)


def test_reductions_gloo(tmp_path):
    _run("_check_reductions", tmp_path)


def test_resample_protocol_gloo(tmp_path):
    _run("_check_resample_protocol", tmp_path)


# ---------------------------------------------------------------------------------------------
# GPU legs: the full sharded SMCUpdater (HIP kernels + protocol).  One GPU box has one device, so
# (i) two processes share it and talk over gloo, (ii) a world-size-1 RCCL group checks the nccl path.
def _check_sharded_updater(comm, rank, world, tmpdir):
    import warnings
    import torch
    import qinfer_amd as qi
    torch.cuda.set_device(0)
    n_local = 60000
    ts = (9 / 8) ** np.arange(50.0)
    rs = np.random.RandomState(0)
    outcomes = (rs.random_sample(50) >= np.cos(0.3 * ts / 2) ** 2).astype(int)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        upd = qi.SMCUpdater(qi.SimplePrecessionModel(), n_local, qi.UniformDistribution([0, 1]),
                            device_rng=True, seed=5, comm=comm)
        assert upd.n_particles == n_local and upd.n_particles_global == n_local * world
        np.testing.assert_allclose(upd.n_ess, n_local * world, rtol=1e-12)
        for k in range(50):
            upd.update(int(outcomes[k]), ts[k:k + 1])
    rec = np.array([upd.resample_count, upd.n_ess, upd.min_n_ess] + list(upd.normalization_record) + list(upd.est_mean()))
    rows = comm.gather_rows(torch.from_numpy(rec))
    for r in range(1, world):
        assert np.array_equal(rows[0], rows[r]), "ranks disagree on the global quantities"
    np.save(os.path.join(tmpdir, "locs_%d.npy" % rank), upd.particle_locations)
    comm.dist.barrier()
    if rank == 0:
        import np_oracle as orc
        np.random.seed(2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = orc.OracleSMC(orc.precession_model(), 20000, lambda m: np.random.random((m, 1)))
            for k in range(50):
                ref.update(int(outcomes[k]), {"t": ts[k:k + 1]})
        sd = np.sqrt(ref.est_covariance_mtx()[0, 0])
        assert abs(upd.est_mean()[0] - ref.est_mean()[0]) < 0.25 * sd
        assert abs(upd.resample_count - ref.resample_count) <= 3
        shards = [np.load(os.path.join(tmpdir, "locs_%d.npy" % r)) for r in range(world)]
        if world > 1:                               # shards are statistically exchangeable
            # (the cloud is the proposal of the last resample: compare with ITS spread, not the posterior's)
            spread = max(s_.std() for s_ in shards)
            assert abs(shards[0].mean() - shards[1].mean()) < 6 * spread * np.sqrt(2.0 / n_local)
            assert abs(shards[0].std() / shards[1].std() - 1) < 0.05
        assert min(s.min() for s in shards) > 0


@pytest.mark.gpu
def test_sharded_updater_two_ranks_one_gpu(tmp_path):
    _run("_check_sharded_updater", tmp_path, world=2)


def _nccl_world1(rank, port, tmpdir):
    for p in (os.path.join(ROOT, "python-qinfer_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from qinfer_amd.parallel import ParticleShardGroup
    comm = ParticleShardGroup(seed=1234)
    assert comm.backend == "nccl"
    _check_sharded_updater(comm, 0, 1, tmpdir)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_updater_rccl_world1(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_nccl_world1, args=(_free_port(), str(tmp_path)), nprocs=1, join=True)
