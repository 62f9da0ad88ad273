"""Resamplers.  `Resampler` ABC and `LiuWestResampler` with the reference's constructor and call
signatures (resamplers.py:73-95, 171-392); the work runs in HIP kernels:

    moments (one fused pass) -> host sqrtm_psd (d x d) -> device scan of the weights ->
    ancestor search + Liu-West centres + Gaussian kick + validity mask -> uniform weights.

Two RNG modes:

* legacy (default, `device_rng=False`): uniforms come from `np.random.random` and normals from
  `kernel(d, k)` on the host, in exactly the order and shapes the reference draws them, and are
  uploaded.  Seed-for-seed comparable with QInfer (incl. quirk Q1: on a postselection retry the
  reference re-uses the FIRST k centres, `mus = mus[:k]`, resamplers.py:371-372).
* device (`device_rng=True`): Philox4x32-10 inside a single kernel launch; an invalid particle
  redraws its ancestor and kick in-thread.  Statistically equivalent, not stream-identical.
"""
import abc
import math
import warnings

import numpy as np

from ._exceptions import ResamplerError, ResamplerWarning
from .abstract_model import native_ok
from .distributions import ParticleDistribution

__all__ = ["Resampler", "LiuWestResampler"]

_MAX_D = 64          # _native.QSMC_MAX_D_WIDE: the widest cloud the device samplers take (above 16: no validity test of their own)


class Resampler(metaclass=abc.ABCMeta):
    @abc.abstractmethod
    def __call__(self, model, particle_dist, n_particles=None, precomputed_mean=None,
                 precomputed_cov=None):
        """Return a resampled ParticleDistribution with `n_particles` particles."""


class LiuWestResampler(Resampler):
    r"""Liu & West (2001) kernel-shrinkage resampler: x_i' ~ N(a x_j + (1-a) mu, h^2 Sigma).

    :param float a: shrinkage; `h` defaults to sqrt(1 - a^2) (moment preserving).
    :param int maxiter: postselection retries before giving up with a ResamplerWarning.
    :param bool postselect: redraw particles the model declares invalid.
    :param float zero_cov_comp: diagonal added when the covariance has exactly zero norm.
    :param callable kernel: host `kernel(d, k)` -> zero-mean unit-variance draws (legacy mode).
    :param int default_n_particles: output cloud size (None = same as the input).
    :param bool device_rng: use the on-device Philox generator (see module docstring).
    :param int seed: Philox key for `device_rng=True`.
    :param bool legacy_mus_truncation: reproduce quirk Q1 in legacy mode (default True).
    """

    _override_h = False

    def __init__(self, a=0.98, h=None, maxiter=1000, debug=False, postselect=True, zero_cov_comp=1e-10,
                 default_n_particles=None, kernel=np.random.randn, device_rng=False, seed=0,
                 legacy_mus_truncation=True):
        self._default_n_particles = default_n_particles
        self.a = a
        if h is not None:
            self._override_h = True
            self._h = h
        self._maxiter = maxiter
        self._debug = debug
        self._postselect = postselect
        self._zero_cov_comp = zero_cov_comp
        self._kernel = kernel
        # (maxiter < 1 -- "draw nothing": the reference leaves every new location at zero and warns, resamplers.py:322-381 --
        #  has no device form, the samplers take at least one round: that degenerate setting runs the host-replay path)
        self._device_rng = bool(device_rng) and int(maxiter) >= 1
        self._seed = int(seed)
        self._epoch = 0
        self._legacy_q1 = bool(legacy_mus_truncation)

    @property
    def a(self):
        return self._a

    @a.setter
    def a(self, new_a):
        self._a = new_a
        if not self._override_h:
            self._h = np.sqrt(1 - new_a ** 2)

    @property
    def h(self):
        return self._h

    # ------------------------------------------------------------------------------------------
    def __call__(self, model, particle_dist, n_particles=None, precomputed_mean=None,
                 precomputed_cov=None):
        if not isinstance(particle_dist, ParticleDistribution):
            raise TypeError("particle_dist must be a qinfer_amd ParticleDistribution")
        eng = particle_dist._eng
        mean = particle_dist.est_mean() if precomputed_mean is None else np.asarray(precomputed_mean, float)
        cov = particle_dist.est_covariance_mtx() if precomputed_cov is None else np.asarray(precomputed_cov, float)
        if n_particles is None:
            n_particles = (particle_dist.n_particles if self._default_n_particles is None
                           else self._default_n_particles)
        n_particles = int(n_particles)
        d = particle_dist.n_rvs
        a, h = self._a, self._h
        # (this call sits between the update that failed the n_ess test and the launch of the sampling kernel, with the
        #  GPU waiting: scalar tests on plain floats where d = 1, cached model facts where the updater has them)
        if not (cov[0, 0] != 0.0 if cov.shape == (1, 1) else cov.any()):   # la.norm(cov, 'fro') == 0  (resamplers.py:283)
            warnings.warn("Covariance has zero norm; adding in small covariance in resampler. "
                          "Consider increasing n_particles to improve covariance estimates.",
                          ResamplerWarning)
            cov = self._zero_cov_comp * np.eye(d)
        hint = particle_dist.__dict__.pop("_queued_sqrt", None)      # (SMCUpdater: qsmc_step has formed it already)
        if hint is not None and hint[1] == float(h) and hint[0] == np.ascontiguousarray(cov, dtype=np.float64).tobytes():
            S, S_err = hint[2], hint[3]
        else:
            S, S_err = eng.sqrtm_psd(cov, scale=h)
        if not math.isfinite(S_err):
            raise ResamplerError("Infinite error in computing the square root of the covariance "
                                 "matrix. Check that n_ess is not too small.")

        if getattr(particle_dist, "model", None) is model and hasattr(particle_dist, "_desc"):
            native, desc = particle_dist._native, particle_dist._desc      # an SMCUpdater and its own model
        else:
            native = native_ok(model)
            desc = model._native_desc() if native else None
        x_in, norm = particle_dist._x, particle_dist._norm

        canon = None
        if self._device_rng and native:
            # straight from the weights: the CDF is scanned chunk-wise inside the sampler, never in HBM
            self._epoch += 1
            defer = bool(getattr(self, "_defer_failed_check", False))
            if particle_dist.n_particles > self._segment_limit:
                x_new, n_failed = self._segmented_resample(eng, desc, x_in, particle_dist._w, a, mean, S, n_particles)
                return ParticleDistribution._from_device(eng, x_new, None, norm=float(n_particles),
                                                         sumsq=float(n_particles))
            self._arm_update_sums(particle_dist)
            # an updater keeps a spare cloud of its own size: the new particles go there (and qsmc_step may have
            # queued this very call already -- same arguments, same buffer: the library then has nothing left to do)
            spare = getattr(particle_dist, "_x_spare", None)
            if spare is not None and (tuple(spare.shape) != (d, n_particles) or spare is x_in):
                spare = None
            # an updater that canonicalizes its cloud after every resample (smc.py:529) may have it done by the
            # resample's own kernels (2-qubit tomography): the returned cloud is marked so that it is not done twice
            canon = getattr(particle_dist, "_fused_canon", None)
            if canon is not None and not eng.fused_canon_applies(d, particle_dist.n_particles, n_particles):
                canon = None
            x_new, n_failed = eng.lw_resample_philox(desc, self._postselect, x_in, particle_dist._w, norm, a,
                                                     mean, S, n_particles, self._seed, self._epoch,
                                                     self._maxiter, sync=not defer, out=spare, canon=canon,
                                                     expect_redraws=self._expected_redraws(particle_dist, eng))
            if defer:                  # stay asynchronous: the count is read at the caller's next sync
                self._pending_failed = eng
                n_failed = 0
        elif self._device_rng and d <= _MAX_D and particle_dist.n_particles <= self._segment_limit:
            # a model without native kernels (a user plugin) under the device generator: ancestors and kicks from the same
            # Philox sampler, no validity test of its own; the MODEL's test -- on the device if it has
            # `are_models_valid_device`, else its NumPy one on a host copy of the new particles -- then redraw rounds
            self._epoch += 1
            x_new, n_failed = self._plugin_device_draw(eng, model, x_in, particle_dist._w, norm, a, mean, S, n_particles,
                                                       self._seed, self._epoch)
        else:
            cdf = eng.cumsum(particle_dist._weights(), norm)
            x_new, n_failed = self._legacy_draw(eng, model, desc, x_in, cdf, a, mean, S, n_particles)
        if n_failed:
            warnings.warn("Liu-West resampling failed to find valid models for {} particles within "
                          "{} iterations.".format(n_failed, self._maxiter), ResamplerWarning)

        # uniform weights np.ones(n) / n (resamplers.py:390), held implicitly: w = None means all-ones
        # with normaliser n, so no fill pass is spent and the next update reads 8 B/particle less
        new = ParticleDistribution._from_device(eng, x_new, None, norm=float(n_particles), sumsq=float(n_particles))
        new._canonicalized = bool(self._device_rng and native and canon is not None)
        return new

    @staticmethod
    def _plugin_valid(eng, model, x):
        """bool device tensor [n]: the model's own validity test of the SoA particles x (d, n) -- its device hook when it
        has one, else `are_models_valid` on a host copy (the plugin contract, abstract_model.py:286-300)."""
        t = eng.torch
        uk = getattr(model, "_qsmc_user_kernel", None)
        if uk is not None and uk.has_valid:               # the model's compiled valid() (likelihood_hip)
            return eng.valid_user(uk, x).to(t.bool)
        fn = getattr(model, "are_models_valid_device", None)
        if fn is not None:
            ok = fn(x)
            if not isinstance(ok, t.Tensor) or tuple(ok.shape) != (x.shape[1],):
                raise TypeError("are_models_valid_device must return a device tensor of shape (n_particles,)")
            return ok.to(device=x.device, dtype=t.bool)
        ok = np.asarray(model.are_models_valid(np.ascontiguousarray(x.cpu().numpy().T)), dtype=bool)
        assert ok.ndim == 1, "are_models_valid returned tensor, expected vector."
        return eng.to_device(ok)

    def _plugin_device_draw(self, eng, model, x_in, w, norm, a, mean, S, n_out, seed, epoch):
        """Liu-West draw of `n_out` particles for a model WITHOUT native kernels, on the device generator.  The Philox sampler
        draws (no validity test of its own), the MODEL tests (`_plugin_valid`), and what it rejects is replaced by rejection
        sampling -- ancestor and kick drawn again, like the native device path:
          * from SPARES drawn in the same sampler call: 1.5 x the rejections this resampler saw last time (+ 1024) extra
            outputs behind the n_out wanted ones -- i.i.d. draws of the same proposal, so the valid ones among them are
            exactly what a redraw would have produced -- one validity pass over all of them, one indexed copy;
          * then, for what is still invalid (the first resample, or more rejections than spares), in rounds of fresh draws
            from Philox streams of their own, up to `maxiter`.
        Returns (x_new, n_failed); x_new is a (d, n_out) view whose row stride may exceed n_out (the spares sit behind it)."""
        from . import _native
        d = x_in.shape[0]
        plain = _native.ModelDesc(_native.MODEL_TOMOGRAPHY, d, 0.0, 1, 0)      # (kind only sizes the sampler: no validity test)
        if not self._postselect or n_out == 0:
            x_new, _ = eng.lw_resample_philox(plain, False, x_in, w, norm, a, mean, S, n_out, seed, epoch, self._maxiter,
                                              sync=False)
            return x_new, 0
        seen = int(getattr(self, "_plugin_bad_seen", 0))
        spares = 0
        if seen:
            # (in coarse steps -- 1/64 of the cloud, at least 65536 -- so that successive resamples ask the allocator for the
            #  same few sizes: a new size per resample is a fresh hipMalloc of the whole cloud each time, milliseconds)
            q = 1 << max(16, (max(n_out, 2) - 1).bit_length() - 6)
            spares = -(-(int(1.5 * seen) + 1024) // q) * q
            # ... and never fewer than before for this output size: the buffer's size only ever steps up
            kept = getattr(self, "_plugin_spares", (0, 0))
            if kept[0] == n_out:
                spares = max(spares, kept[1])
            self._plugin_spares = (n_out, spares)
        spares += (n_out + spares) & 1                                          # (an even row stride: 16-byte loads downstream)
        x_all, _ = eng.lw_resample_philox(plain, False, x_in, w, norm, a, mean, S, n_out + spares, seed, epoch, self._maxiter,
                                          sync=False)
        ok = self._plugin_valid(eng, model, x_all)
        bad = (~ok[:n_out]).nonzero(as_tuple=False).reshape(-1)
        self._plugin_bad_seen = int(bad.numel())
        x_new = x_all[:, :n_out] if spares else x_all
        if bad.numel() and spares:
            good = ok[n_out:].nonzero(as_tuple=False).reshape(-1)[:bad.numel()] + n_out
            k = int(good.numel())
            if k:
                x_all[:, bad[:k]] = x_all[:, good]
                bad = bad[k:]
        rounds = 1
        while bad.numel() and rounds < self._maxiter:
            k = int(bad.numel())
            x_r, _ = eng.lw_resample_philox(plain, False, x_in, w, norm, a, mean, S, k,
                                            seed ^ (0xD1B54A32D192ED03 * rounds & (2 ** 64 - 1)), epoch, self._maxiter,
                                            sync=False)
            x_new[:, bad] = x_r
            bad = bad[~self._plugin_valid(eng, model, x_r)]
            rounds += 1
        return x_new, int(bad.numel())

    def _expected_redraws(self, particle_dist, eng):
        """How many first tries of the previous resample of THIS cloud failed postselection: what the library is told to
        bank spare proposals for (qsmc_lw_expect_redraws).  An updater on the per-datum C path keeps the number in its
        qsmc_step_t; otherwise this resampler reads it back when it is next called (it is published with the sums of
        the update that followed the resample)."""
        st = getattr(particle_dist, "_st", None)
        if st is not None:
            return int(st.lw.redraws_seen)
        if getattr(self, "_redraw_pending", False):
            self._redraws_seen = int(eng.last_resample_redraws())
        self._redraw_pending = True
        return int(getattr(self, "_redraws_seen", 0))

    @staticmethod
    def _arm_update_sums(particle_dist):
        """If the weights are the untouched output of the engine's latest fused update, let the resampler start
        from that kernel's per-tile sums instead of re-reading the weights (qsmc_lw_use_update_sums)."""
        tok = getattr(particle_dist, "_w_token", 0)
        if tok and tok == particle_dist._eng.update_gen:
            particle_dist._eng.use_update_sums(tok)

    def _prepare_device(self, model, particle_dist):
        """Called by SMCUpdater the moment its n_ess test fails: queue the part of the resample that needs
        only the weights (chunk sums, multinomial chunk counts, work-item plan) so that the GPU is busy
        while `__call__` forms mean / covariance / sqrtm on the host.  Same arguments as the
        `lw_resample_philox` call that follows, which then starts at the sampling kernel."""
        if not (self._device_rng and native_ok(model)):
            return
        if particle_dist.n_particles > self._segment_limit:
            return                                   # segmented resample: every segment runs its own prefix
        n_out = (particle_dist.n_particles if self._default_n_particles is None else self._default_n_particles)
        self._arm_update_sums(particle_dist)
        particle_dist._eng.lw_resample_prepare(particle_dist._w, particle_dist.n_particles, particle_dist._norm,
                                               int(n_out), self._seed, self._epoch + 1)

    # The LDS-bucketed sampler handles clouds of up to 8192 chunks of 4096 particles.  A larger cloud is resampled
    # as a handful of contiguous segments, exactly like shards on several GPUs (parallel.py): the number of
    # children per segment is one multinomial draw over the segment weights (host Philox, keyed by seed and
    # epoch), then every segment draws its children from its own weights -- the same joint law as one global
    # multinomial -- into its slice of the new cloud.
    _segment_limit = 8192 * 4096

    def _segmented_resample(self, eng, desc, x_in, w, a, mean, S, n_out):
        n_in = x_in.shape[1]
        n_seg = -(-n_in // self._segment_limit)
        seg = -(-n_in // n_seg)
        seg = -(-seg // 4096) * 4096                                  # whole chunks: 16-byte aligned slices
        bounds = [(g * seg, min(n_in, (g + 1) * seg)) for g in range(n_seg) if g * seg < n_in]
        if w is None:
            W = np.array([b - a0 for a0, b in bounds], dtype=np.float64)
        else:
            W = np.array([eng.weight_stats(w[a0:b], 1.0).sum for a0, b in bounds])
        gen = np.random.Generator(np.random.Philox(key=self._seed & (2 ** 64 - 1), counter=[int(self._epoch), 2, 0, 0]))
        T = gen.multinomial(int(n_out), W / W.sum())
        x_new = eng.empty(x_in.shape[0], n_out)
        off, n_failed = 0, 0
        for g, (a0, b) in enumerate(bounds):
            t_g = int(T[g])
            if t_g == 0:
                continue
            _, f = eng.lw_resample_philox(desc, self._postselect, x_in[:, a0:b], None if w is None else w[a0:b],
                                          float(W[g]), a, mean, S, t_g, self._seed + 0x9E3779B97F4A7C15 * (g + 1),
                                          self._epoch, self._maxiter, sync=True, out=x_new[:, off:off + t_g])
            n_failed += f
            off += t_g
        if n_failed:
            warnings.warn("Liu-West resampling failed to find valid models for {} particles within "
                          "{} iterations.".format(n_failed, self._maxiter), ResamplerWarning)
        return x_new, 0

    def _flush_failed_warning(self, synchronize=False):
        """Emit the deferred 'failed to find valid models' ResamplerWarning, if one is due."""
        eng = getattr(self, "_pending_failed", None)
        if eng is None:
            return
        self._pending_failed = None
        n_failed = eng.last_resample_failed(synchronize)
        if n_failed:
            warnings.warn("Liu-West resampling failed to find valid models for {} particles within "
                          "{} iterations.".format(n_failed, self._maxiter), ResamplerWarning)

    # ------------------------------------------------------------------------------------------
    def _legacy_draw(self, eng, model, desc, x_in, cdf, a, mean, S, n_out):
        """Host-RNG path: same draw order / shapes as resamplers.py:318-372."""
        from . import _native
        t = eng.torch
        d = x_in.shape[0]
        u = eng.to_device(np.random.random((n_out,)))
        js = eng.lw_ancestors(cdf, u)
        mus = eng.lw_centres(x_in, js, a, mean)
        x_new = eng.empty(d, n_out)
        kernel_desc = desc if desc is not None else _native.ModelDesc(_native.MODEL_TOMOGRAPHY, d, 0.0, 1, 0)
        device_valid = desc is not None
        idxs = None                      # None = identity (first round)
        k = n_out
        n_iters = 0
        while k and n_iters < self._maxiter:
            n_iters += 1
            z = eng.to_device(np.asarray(self._kernel(d, k), dtype=np.float64).reshape(d, k))
            centre_by_idx = (idxs is not None) and (not self._legacy_q1)
            valid = eng.lw_perturb(kernel_desc, self._postselect and device_valid, mus, idxs, k,
                                   centre_by_idx, S, z, x_new)
            if self._postselect and not device_valid:
                # plugin slow path: the user's are_models_valid decides, on a host copy
                cols = x_new if idxs is None else x_new[:, idxs]
                ok = np.asarray(model.are_models_valid(np.ascontiguousarray(cols.cpu().numpy().T)), dtype=bool)
                assert ok.ndim == 1, "are_models_valid returned tensor, expected vector."
                valid = eng.to_device(ok.astype(np.uint8))
            bad = (valid == 0).nonzero(as_tuple=False).reshape(-1)      # ascending, like np.nonzero
            if bad.numel() == 0:
                k = 0
                break
            idxs = bad if idxs is None else idxs[bad]
            idxs = idxs.contiguous()
            k = int(idxs.shape[0])
        return x_new, k
