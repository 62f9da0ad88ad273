"""Particle sharding across the GPUs of a node: one process per GPU, `torch.distributed`
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).

The reference's only data-parallel mechanism is `DirectViewParallelizedModel` (parallel.py:76-288):
split the particle axis over ipyparallel engines for the likelihood call, gather `L` back, keep the
weights and the resampler on the client.  Here every rank OWNS a contiguous block of N/G particles
(locations and weights never leave its HBM), and only O(1)-sized reductions cross xGMI:

  per datum      all-gather of 4 doubles/rank  [sum w', sum w'^2, min w', #bad]  -> every rank forms
                 the same global normaliser, n_ess and resample decision (summed in rank order, so
                 bitwise identical everywhere);
  per resample   all-gather of 1 + d + d(d+1)/2 doubles/rank (weighted moments) -> identical global
                 mean/cov -> identical host sqrtm on every rank; then the only bandwidth step:
                 an all-to-all(v) of ancestor rows (8 d bytes each).

Exact global multinomial resampling without a global CDF: ancestors for destination rank r come
from source rank h with probability W_h / W, so the G x G count matrix C[r, h] ~ Multinomial(N/G;
W/W_tot) is drawn IDENTICALLY on every rank from a shared-seed host generator; rank h then draws
C[., h] ancestors from its LOCAL CDF (conditionally i.i.d. -- exact) with the same LDS-bucketed
sampler the single-GPU path uses, applies the Liu-West kick and postselection there (they need only
the ancestor and the global mean/covariance), and the finished rows travel by one
`all_to_all_single`.  xGMI is point-to-point, so the all-to-all is a direct exchange (worst case 7/8
of the rows leave the rank), not a ring.

The collectives take whatever tensors they are given (CUDA under nccl, CPU under gloo), so the
protocol is testable on CPU with world_size 2 (tests/test_parallel_gloo.py).
"""
import numpy as np

__all__ = ["ParticleShardGroup"]


class ParticleShardGroup:
    def __init__(self, group=None, seed=0):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised (launch with torch.distributed.run)")
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.seed = int(seed)
        self._epoch = 0

    # ------------------------------------------------------------------ small collectives
    def _comm_tensor(self, t):
        """Tensor on the device this backend communicates from."""
        if self.backend == "nccl":
            return t if t.is_cuda else t.cuda()
        return t.cpu() if t.is_cuda else t

    def gather_rows(self, vec):
        """All-gather a 1-D float64 tensor: returns a HOST (world, len) ndarray, rank-ordered."""
        t = self.torch
        v = self._comm_tensor(vec.reshape(-1).to(t.float64)).contiguous()
        out = t.empty(self.world_size * v.shape[0], dtype=t.float64, device=v.device)
        self.dist.all_gather_into_tensor(out, v, group=self.group)       # flat: accepted by nccl and gloo
        return out.cpu().numpy().reshape(self.world_size, v.shape[0])

    def combine_update_stats(self, stats):
        """stats: tensor [sum, sumsq, min, n_bad, extra...] of this shard -> global
        (sum, sumsq, min, n_bad).  Extra entries (e.g. the fused moment sums) are summed too and kept,
        with the per-rank weight sums, in `last_extra` / `last_shard_sums`: ONE collective per datum
        serves the normaliser, n_ess, the guards, est_mean/est_covariance and the next resample plan."""
        rows = self.gather_rows(stats)
        tot = np.zeros(rows.shape[1])
        for r in range(self.world_size):            # fixed order: identical on every rank
            tot += rows[r]
        self.last_shard_sums = rows[:, 0].copy()
        self.last_extra = tot[4:].copy()
        return float(tot[0]), float(tot[1]), float(rows[:, 2].min()), float(tot[3])

    def allreduce_update_stats(self, eng, s, ss, mn, n_bad, extra=None):
        t = self.torch
        vec = [s, ss, mn, n_bad] + ([] if extra is None else [float(v) for v in extra])
        return self.combine_update_stats(t.tensor(vec, dtype=t.float64))

    def allreduce_scalar(self, eng, value):
        t = self.torch
        rows = self.gather_rows(t.tensor([value], dtype=t.float64))
        tot = 0.0
        for r in range(self.world_size):
            tot += rows[r, 0]
        return float(tot)

    def allreduce_moments(self, eng, s0, s1, s2):
        """Global [sum w, sum w x, sum w x x^T] from per-shard sums (already divided by the GLOBAL norm)."""
        t = self.torch
        d = len(s1)
        packed = np.concatenate([[s0], s1, s2.reshape(-1)])
        rows = self.gather_rows(t.from_numpy(packed))
        tot = np.zeros_like(packed)
        for r in range(self.world_size):
            tot += rows[r]
        return float(tot[0]), tot[1:1 + d].copy(), tot[1 + d:].reshape(d, d).copy()

    def allreduce_tensor(self, tensor):
        v = self._comm_tensor(tensor).contiguous()
        self.dist.all_reduce(v, op=self.dist.ReduceOp.SUM, group=self.group)
        return v.to(tensor.device)

    # ------------------------------------------------------------------ resampling protocol
    def plan_counts(self, shard_weights, n_out_per_rank, epoch):
        """C[r, h] = how many ancestors destination r takes from source h; identical on all ranks."""
        p = np.asarray(shard_weights, dtype=np.float64)
        p = p / p.sum()
        gen = np.random.Generator(np.random.Philox(key=self.seed & (2 ** 64 - 1), counter=[int(epoch), 0, 0, 0]))
        return gen.multinomial(int(n_out_per_rank), p, size=self.world_size).astype(np.int64)

    def exchange_rows(self, send_rows, counts):
        """send_rows: (T_h, d) tensor ordered by destination rank, T_h = counts[:, rank].sum().
        Returns the (n_local, d) rows this rank receives, ordered by source rank."""
        t = self.torch
        send_split = [int(c) for c in counts[:, self.rank]]
        recv_split = [int(c) for c in counts[self.rank, :]]
        v = self._comm_tensor(send_rows).contiguous()
        assert v.shape[0] == sum(send_split)
        out = t.empty((sum(recv_split), v.shape[1]), dtype=v.dtype, device=v.device)
        self.dist.all_to_all_single(out, v, output_split_sizes=recv_split, input_split_sizes=send_split,
                                    group=self.group)
        return out

    def resample(self, updater, resampler):
        """Sharded Liu-West step for `updater` (an SMCUpdater with comm=self).  Device RNG only."""
        import warnings
        from ._exceptions import ResamplerError, ResamplerWarning
        from .distributions import ParticleDistribution
        eng = updater._eng
        model = updater.model
        if not getattr(model, "_native", False):
            raise NotImplementedError("sharded resampling needs a model with native kernels")
        self._epoch += 1
        epoch = self._epoch
        d = updater.n_rvs
        n_local = updater.n_particles
        mean = updater.est_mean()                                   # global (all-reduced) moments
        cov = updater.est_covariance_mtx()
        a, h = resampler.a, resampler.h
        if not cov.any():
            warnings.warn("Covariance has zero norm; adding in small covariance in resampler. "
                          "Consider increasing n_particles to improve covariance estimates.", ResamplerWarning)
            cov = resampler._zero_cov_comp * np.eye(d)
        S, S_err = eng.sqrtm_psd(cov, scale=h)
        if not np.isfinite(S_err):
            raise ResamplerError("Infinite error in computing the square root of the covariance "
                                 "matrix. Check that n_ess is not too small.")
        # shard weight totals (unnormalised sums are fine: only ratios matter); normally known from the
        # last update's all-gather, so the resample needs no collective besides the all-to-all
        W = getattr(updater, "_shard_sums", None)
        if W is None:
            st = eng.weight_stats(updater._weights(), 1.0)
            W = self.gather_rows(self.torch.tensor([st.sum], dtype=self.torch.float64))[:, 0]
        counts = self.plan_counts(W, n_local, epoch)
        seed_r = resampler._seed + 0x9E3779B97F4A7C15 * (self.rank + 1)
        # this shard draws, kicks and postselects the particles every destination takes from it
        defer = hasattr(resampler, "_flush_failed_warning")       # stay asynchronous; warn at the next sync
        rows, n_failed = eng.lw_resample_philox_sharded(model._native_desc(), resampler._postselect, updater._x,
                                                        updater._w, float(W[self.rank]),   # local normaliser
                                                        a, mean, S, counts[:, self.rank], seed_r, epoch,
                                                        resampler._maxiter, sync=not defer)
        if defer:
            resampler._pending_failed = eng
        recv = self.exchange_rows(rows, counts)                      # (n_local, d), the only bandwidth step
        x_new = recv.to(eng.device).t().contiguous()                 # back to SoA
        if n_failed:
            warnings.warn("Liu-West resampling failed to find valid models for {} particles within {} "
                          "iterations.".format(n_failed, resampler._maxiter), ResamplerWarning)
        n_total = n_local * self.world_size
        return ParticleDistribution._from_device(eng, x_new, None, norm=float(n_total), sumsq=float(n_total))
