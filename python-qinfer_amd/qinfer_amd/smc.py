"""SMCUpdater: sequential Monte Carlo Bayes updates on a GPU-resident particle cloud.

Drop-in for the reference's `qinfer.smc.SMCUpdater` (smc.py:97-551): same constructor, methods,
properties, warnings and exceptions.  Differences are under the hood:

* one fused kernel per datum: likelihood x weight multiply x (sum, sum of squares, min, #bad)
  reductions -- the reference does ~25 NumPy passes for the same thing;
* weights live unnormalised in HBM (`true w = w / norm`); every quantity the reference derives
  from normalised weights (norm record, n_ess, zero-weight test) comes from the in-kernel sums;
* `particle_locations` / `particle_weights` are properties returning NumPy copies -- mutate by
  assigning a whole array.

Extension kwargs (all default to reference behaviour): `device_rng`/`seed` switch prior sampling
and the default resampler to the on-device Philox generator; `comm` shards the cloud across ranks
(see parallel.py).
"""
import warnings

import numpy as np

from . import _native
from ._exceptions import ApproximationWarning, ResamplerWarning
from .abstract_model import Simulatable, native_ok
from .distributions import ParticleDistribution
from .resamplers import LiuWestResampler

__all__ = ["SMCUpdater"]

_EPS = float(np.spacing(1))
# test hooks (module attributes the parity tests patch; never read from the environment by the product):
_NO_STEP = False       # True: `update` takes the Python per-datum path instead of qsmc_step -- the independent form of the C path
_NO_ADOPT = False      # True: the resampler's own call re-derives a resample qsmc_step has queued
_NO_FUSED_CANON = False   # True: canonicalize after a d = 16 resample as its own passes, not inside the kick kernel
_U64 = 2 ** 64 - 1


def _user_ep_matrix(expparams):
    """(n_experiments, n_ep) float64: an experiment record as the doubles a compiled user model reads (`likelihood_hip`:
    the fields in dtype order, vector fields flattened; a plain-dtype array is one double per experiment)."""
    ep = np.atleast_1d(expparams)
    if ep.dtype.names is None:
        return np.ascontiguousarray(ep.astype(np.float64).reshape(len(ep), -1))
    cols = [np.asarray(ep[name], dtype=np.float64).reshape(len(ep), -1) for name in ep.dtype.names]
    return np.ascontiguousarray(np.hstack(cols))


def _as_int_outcome(outcome):
    if type(outcome) is int:
        return outcome
    arr = np.asarray(outcome)
    if arr.size != 1:
        raise ValueError("update() takes a single outcome; use batch_update for several")
    return int(arr.reshape(-1)[0])


_FROM_STEP = ("moments in the qsmc_step_t",)


class SMCUpdater(ParticleDistribution):
    def __init__(self, model, n_particles, prior, resample_a=None, resampler=None, resample_thresh=0.5,
                 debug_resampling=False, track_resampling_divergence=False, zero_weight_policy='error',
                 zero_weight_thresh=None, canonicalize=True, device_rng=False, seed=0, comm=None):
        from .engine import get_engine
        self._eng = get_engine()
        self._comm = comm
        self._moments_cache = None
        self._x = self._eng.empty(model.n_modelparams, 0)
        self._w = self._eng.empty(0)
        self._w_alt = None
        self._norm, self._sumsq = 1.0, None

        self._resample_count = 0
        # sharded: a rank passes ITS share; the shares need not be equal (np.array_split's remainders, as the reference's
        # DirectViewParallelizedModel splits a cloud: parallel.py:216-224) -- the ranks' counts are gathered once per
        # explicit size (here and in reset(n)), their sum is the global count every rank works with
        self._shard_counts = None
        self._min_n_ess = n_particles                      # (sharded: the global count, once reset() below has gathered the shares)
        self.model = model
        self.prior = prior
        self._canonicalize = bool(canonicalize)
        self._device_rng = bool(device_rng)
        self._seed = int(seed)
        self._prior_epoch = 0

        self._debug_resampling = debug_resampling
        if resample_a is not None:
            warnings.warn("The 'resample_a' keyword argument is deprecated; use "
                          "'resampler=LiuWestResampler(a)' instead.", DeprecationWarning)
            if resampler is not None:
                raise ValueError("Both a resample_a and an explicit resampler were provided; please "
                                 "provide only one.")
            self.resampler = LiuWestResampler(a=resample_a, device_rng=device_rng, seed=seed)
        elif resampler is None:
            self.resampler = LiuWestResampler(default_n_particles=n_particles, device_rng=device_rng,
                                              seed=seed)
        else:
            self.resampler = resampler
        self.resample_thresh = resample_thresh

        self._just_resampled = False
        self._data_record = []
        self._normalization_record = []
        self._resampling_divergences = [] if track_resampling_divergence else None
        if track_resampling_divergence and comm is not None:
            warnings.warn("track_resampling_divergence compares whole clouds (an O(N^2) kernel-density estimate); "
                          "a sharded updater holds only its shard: divergences will not be recorded.",
                          ApproximationWarning)
            self._resampling_divergences = None
        self._zero_weight_policy = zero_weight_policy
        self._zero_weight_thresh = (zero_weight_thresh if zero_weight_thresh is not None
                                    else 10 * np.spacing(1))
        # the HIP kernels stand for the model only if its kernel-backed methods are the library's own
        # (a user subclass overriding likelihood / are_models_valid / update_timestep takes the plugin path)
        self._native = native_ok(model)
        self._desc = model._native_desc() if self._native else None
        # a user model that states its likelihood as HIP source (`likelihood_hip`): compiled into the fused update kernel now
        self._uk = None
        src = None if self._native else getattr(model, "likelihood_hip", None)
        if src:
            n_ep = _user_ep_matrix(np.zeros((1,), dtype=model.expparams_dtype)).shape[1]
            self._uk = self._eng.user_kernel(src, model.n_modelparams, n_ep)
            model._qsmc_user_kernel = self._uk          # (the resampler's postselection asks the model: _plugin_valid)
        self._timestep_identity = (self._timestep_is_identity(model)
                                   and (self._native or getattr(model, "update_timestep_device", None) is None))
        # the per-datum C path (qsmc_step): native model; one cloud, or a shard whose per-datum reduction goes through
        # shared memory (the C call then makes that collective too)
        self._st_exchange = None if comm is None else comm.step_exchange()
        self._st = _native.Step() if (self._native and not _NO_STEP
                                      and (comm is None or self._st_exchange is not None)) else None
        if self._st is not None:
            import ctypes
            self._st_ref = ctypes.byref(self._st)
            self._desc_ref = ctypes.byref(self._desc)
            self._ep = _native.ExpParam()
            self._ep_ref = ctypes.byref(self._ep)
            self._ep_fill = getattr(model, "_native_fill_expparam", None)
            # NumPy views of the struct's result arrays (no copy; a ctypes slice builds a Python list first: 256 floats
            # per matrix, on the resample path of every d = 16 step while the GPU waits)
            as_arr = np.ctypeslib.as_array
            self._st_cov, self._st_S = as_arr(self._st.cov), as_arr(self._st.S)
            self._st_mom, self._st_mom_big = as_arr(self._st.moments), as_arr(self._st.moments_big)
        self._step_synced = False
        self._x_spare = None
        # canonicalize after a resample (smc.py:529) done by the resample's own kernels where the library can
        fc = getattr(model, "_native_canonicalize_fused", None)
        self._fused_canon = (fc(self._eng) if (fc is not None and self._native and self._canonicalize and not _NO_FUSED_CANON)
                             else None)
        self.reset(n_particles)
        if comm is not None:
            self._min_n_ess = self._n_global

    # ------------------------------------------------------------------ bookkeeping properties
    @property
    def resample_count(self):
        return self._resample_count

    @property
    def just_resampled(self):
        return self._just_resampled

    @property
    def normalization_record(self):
        return self._normalization_record

    @property
    def log_total_likelihood(self):
        return np.sum(np.log(self.normalization_record))

    @property
    def min_n_ess(self):
        return self._min_n_ess

    @property
    def data_record(self):
        return self._data_record[:]

    @property
    def resampling_divergences(self):
        return self._resampling_divergences

    @property
    def n_particles_global(self):
        """Particles over all shards (== n_particles without a comm).  Constant between resets: under
        local placement the shard sizes float, their sum does not."""
        return self.n_particles if self._comm is None else self._n_global

    # ------------------------------------------------------------------ sharding hooks
    def _gather_shard_counts(self, n_local):
        """Every rank's nominal share, rank-ordered (one small all-gather); cached until the next explicit size."""
        rows = np.asarray(self._comm.gather_rows(np.array([float(n_local)])))[:, 0]
        if not np.all((rows >= 1) & (rows == np.floor(rows))):
            raise ValueError("sharded SMCUpdater: every rank needs at least one particle, got %s" % rows.tolist())
        self._shard_counts = rows.astype(np.int64)
        return self._shard_counts

    def _reduce_stats(self, st, extra=None):
        """Combine per-shard update sums across ranks (one all-gather per datum; see parallel.py)."""
        if self._comm is None:
            return st.sum, st.sumsq, st.min, st.n_bad
        out = self._comm.allreduce_update_stats(self._eng, st.sum, st.sumsq, st.min, st.n_bad, extra)
        self._shard_sums = self._comm.last_shard_sums
        return out

    def _moments(self):
        c = self._moments_cache
        if c is _FROM_STEP:
            # left by the C per-datum path: the packed sums are still in the qsmc_step_t (the next update replaces
            # both them and this marker)
            d = self._x.shape[0]
            raw = self._st_mom[:d + d * (d + 1) // 2].copy()
            c = self._moments_cache = ("packed", 1.0, raw, self._norm)
        if c is not None and len(c) == 4:
            # left by update(): the packed sums [sum w'x, upper(sum w'xx^T)] of the fused kernel and their
            # normaliser -- unpacked only when somebody asks (a resample, est_mean, ...), not on every datum
            _, s0, raw, nrm = c
            d = self._x.shape[0]
            c = self._moments_cache = (s0, raw[:d] / nrm, self._eng._unpack_upper(raw[d:], d) / nrm)
        if self._moments_cache is None:
            if self._comm is None:
                self._moments_cache = self._eng.moments(self._x, self._weights(), self._norm)
            else:
                self._moments_cache = self._comm.allreduce_moments(
                    self._eng, *self._eng.moments(self._x, self._weights(), self._norm))
        return self._moments_cache

    @property
    def n_ess(self):
        if self._sumsq is None:
            st = self._eng.weight_stats(self._weights(), self._norm)
            sumsq = st.sumsq * self._norm * self._norm
            if self._comm is not None:
                sumsq = self._comm.allreduce_scalar(self._eng, sumsq)
            self._sumsq = sumsq
        return self._ess_from(self._sumsq)

    # ------------------------------------------------------------------ initialisation
    def reset(self, n_particles=None, only_params=None, reset_weights=True):
        """Redraw locations (and weights) from the prior (smc.py:281-320)."""
        if n_particles is not None and only_params is not None:
            raise ValueError("Cannot set both n_particles and only_params.")
        if n_particles is None:
            # sharded: the nominal per-rank size, not the current (floating) one, so that every rank
            # agrees on the global count
            n_particles = self.n_particles if self._comm is None else self._n_local_nominal
        elif self._comm is not None:
            self._gather_shard_counts(n_particles)              # (a collective: every rank resets with its own size)
        self._n_local_nominal = n_particles
        eng = self._eng
        d = self.model.n_modelparams
        n_total = n_particles if self._comm is None else int(self._shard_counts.sum())
        self._n_global = n_total
        if reset_weights:
            # uniform weights 1/N (smc.py:307), held implicitly: all-ones with normaliser N
            self._w = None
            self._w_alt = None
            self._norm = float(n_total)
            self._sumsq = float(n_particles) if self._comm is None else float(n_total)
            self._shard_sums = None if self._comm is None else self._shard_counts.astype(np.float64)
        x_new = None
        if self._device_rng and hasattr(self.prior, "sample_device"):
            try:
                self._prior_epoch += 1
                rank = 0 if self._comm is None else self._comm.rank
                x_new, _ = self.prior.sample_device(eng, n_particles, self._seed + 0x9E3779B97F4A7C15 * (rank + 1),
                                                    self._prior_epoch)
            except NotImplementedError:
                x_new = None
        if x_new is None:
            x_new = eng.locs_to_soa(np.asarray(self.prior.sample(n=n_particles), dtype=np.float64))
        if only_params is None:
            self._x = x_new
            rows = slice(None)
        else:
            rows = only_params
            self._x[rows, :] = x_new[rows, :]
        self._invalidate()
        if getattr(self, "_st", None) is not None:
            self._st.lw.redraws_seen, self._st.lw.redraw_pending = 0, 0
        self._shard_resampled = False
        if self._canonicalize:
            self._canonicalize_device(rows)
        r = self.resampler
        if self._native and getattr(r, "_device_rng", False) and n_particles <= getattr(r, "_segment_limit", 0):
            # the scratch a resample of this cloud will want, grown now rather than inside the first resample
            n_out = getattr(r, "_default_n_particles", None)
            eng.reserve(n_particles, n_particles if n_out is None else int(n_out), d)

    @staticmethod
    def _canonicalize_is_identity(model):
        from .models import DerivedModel
        m = model
        while m is not None:
            fn = type(m).canonicalize
            if fn is Simulatable.canonicalize:
                return True
            if fn is DerivedModel.canonicalize:
                m = m.underlying_model
                continue
            return False
        return True

    @staticmethod
    def _timestep_is_identity(model):
        """True if `update_timestep` of the model (through any chain of decorators that merely forward it)
        is the static-parameter default."""
        from .models import BinomialModel, DerivedModel
        m = model
        while m is not None:
            fn = type(m).update_timestep
            if fn is Simulatable.update_timestep:
                return True
            if fn is DerivedModel.update_timestep or fn is BinomialModel.update_timestep:
                m = m.underlying_model
                continue
            return False
        return True

    def _canonicalize_device(self, rows=slice(None)):
        model = self.model
        if self._canonicalize_is_identity(model) and (native_ok(model) or getattr(model, "canonicalize_device", None) is None):
            return                                    # every hot-path model except tomography
        whole = isinstance(rows, slice) and rows == slice(None)
        dev_fn = None if native_ok(model) else getattr(model, "canonicalize_device", None)
        if native_ok(model) and whole and getattr(model, "_native_canonicalize_ok", lambda: False)():
            model._native_canonicalize_(self._eng, self._x)
        elif dev_fn is not None and whole:            # device-side plugin hook: the cloud never leaves HBM
            self._x = self._check_cloud(dev_fn(self._x), "canonicalize_device")
        else:                                         # plugin slow path (plain host arrays: no write-through uploads)
            locs = self._host_locations(private=True)
            locs[:, rows] = model.canonicalize(locs[:, rows])
            self._x = self._eng.locs_to_soa(locs)
        self._invalidate()

    def _host_locations(self, private=False):
        """(N, d) bare ndarray copy of the cloud for the library's own host paths and for user model callbacks
        (`likelihood`, `canonicalize`, `update_timestep`): an in-place operation inside a plugin must not trigger a
        whole-cloud upload per statement, as it would on the write-through snapshot `particle_locations` returns.

        The copy is kept until the locations change (`_invalidate`): an update moves weights, not particles, so a
        NumPy-plugin model pays the 8 d N-byte D2H + transpose once per resample, not once per datum.  The kept array is
        read-only (a plugin that wrote into its `modelparams` argument would silently fork it from the device cloud);
        `private=True` returns a writable copy the caller owns."""
        locs = self._host_locs
        if locs is None:
            locs = np.ascontiguousarray(self._x.cpu().numpy().T)
            locs.flags.writeable = False
            self._host_locs = locs
        return locs.copy() if private else locs

    def _check_cloud(self, x_new, what):
        """A device-side plugin hook's returned cloud: float64 (d, N) on this device, made contiguous."""
        t = self._eng.torch
        if not isinstance(x_new, t.Tensor) or x_new.dtype != t.float64 or x_new.device != self._x.device \
                or tuple(x_new.shape) != tuple(self._x.shape):
            raise TypeError("{} must return a float64 tensor of shape {} on {}".format(
                what, tuple(self._x.shape), self._x.device))
        return x_new.contiguous()

    # ------------------------------------------------------------------ updates
    def hypothetical_update(self, outcomes, expparams, return_likelihood=False, return_normalization=False):
        """Posterior weights for hypothetical data: arrays of shape (n_outcomes, n_expparams,
        n_particles) on the HOST, as in smc.py:324-386."""
        if not isinstance(outcomes, np.ndarray):
            outcomes = np.array([outcomes])
        eng = self._eng
        L = self._device_likelihood(outcomes, expparams)               # (n_o, n_e, N) device
        n_o, n_e, n = L.shape
        w = self._weights()
        hyp = eng.empty(n_o, n_e, n)
        norm_scale = np.empty((n_o, n_e, 1))
        # per (outcome, experiment): hyp = (w / norm) * L with its sum in the same pass (the update kernel's
        # plugin form), then one scaling pass by the zero-fixed normaliser (smc.py:355-372)
        for o in range(n_o):
            for e in range(n_e):
                st = eng.update_from_likelihood(L[o, e], w, hyp[o, e], self._norm)
                norm_scale[o, e, 0] = st.sum
        if self._comm is not None:
            norm_scale = self._comm.allreduce_tensor(eng.torch.from_numpy(norm_scale)).cpu().numpy()
        fixed = np.where(np.abs(norm_scale) < _EPS, 1.0, norm_scale)
        for o in range(n_o):
            for e in range(n_e):
                if fixed[o, e, 0] != 1.0:
                    eng.normalize_weights_into(hyp[o, e], hyp[o, e], fixed[o, e, 0])
        out = [hyp.cpu().numpy()]
        if return_likelihood:
            out.append(L.cpu().numpy())
        if return_normalization:
            out.append(norm_scale)
        return out[0] if len(out) == 1 else tuple(out)

    def _device_likelihood(self, outcomes, expparams):
        outcomes = np.atleast_1d(np.asarray(outcomes)).ravel()
        if self._native:
            return self._eng.likelihood(self._desc, self._x, self.model._native_expparams(expparams),
                                        outcomes.astype(np.int64))
        if self._uk is not None:
            # the model's own HIP source, compiled (kernels/user_jit.hpp): L on the device, nothing crosses the bus
            if hasattr(self.model, "count_likelihood_calls"):
                self.model.count_likelihood_calls(len(outcomes), self._x.shape[1], np.atleast_1d(expparams).shape[0])
            return self._eng.likelihood_user(self._uk, self._x, _user_ep_matrix(expparams), outcomes.astype(np.int64))
        dev_fn = getattr(self.model, "likelihood_device", None)
        if dev_fn is not None:
            # device-side plugin hook: the model evaluates its likelihood on the (d, N) tensor the cloud lives in -- no
            # host copy of the cloud, no upload of L
            t = self._eng.torch
            L = dev_fn(outcomes, self._x, expparams)
            n_e = int(np.shape(expparams)[0]) if np.ndim(expparams) else 1
            if (not isinstance(L, t.Tensor) or L.dtype != t.float64 or L.device != self._x.device
                    or tuple(L.shape) != (len(outcomes), n_e, self._x.shape[1])):
                raise TypeError("likelihood_device must return a float64 tensor of shape (n_outcomes, n_experiments, "
                                "n_particles) = {} on {}".format((len(outcomes), n_e, self._x.shape[1]), self._x.device))
            return L.contiguous()
        L = np.asarray(self.model.likelihood(outcomes, self._host_locations(), expparams), dtype=np.float64)
        return self._eng.to_device(np.ascontiguousarray(L.transpose(0, 2, 1)))

    # ------------------------------------------------------------------ per-datum C path
    def _step_sync(self):
        """Refill the qsmc_step_t from the Python-side state (after a reset, a resample, a user write, a changed
        resampler or threshold -- anything that went through `_invalidate`); nothing here runs per datum."""
        st, x, r = self._st, self._x, self.resampler
        d, n = x.shape
        st.x, st.ldx, st.n = x.data_ptr(), x.stride(0), n
        st.w = self._w.data_ptr() if self._w is not None else None
        st.w_alt = self._scratch_weights().data_ptr()
        st.norm = self._norm
        st.min_n_ess = float(self._min_n_ess)
        st.zero_weight_thresh = float(self._zero_weight_thresh)
        st.ess_below = self.n_particles_global * self.resample_thresh
        lw = st.lw
        key = self._prefix_key()
        ex = self._st_exchange
        if ex is not None:
            # a shard: qsmc_step makes the per-datum reduction over the shards itself; the n_ess test needs every shard's
            # sums and the resample the shard plan, so nothing is armed or queued from C (parallel.py: resample)
            key = None
            if st.ex_segment is None:
                st.ex_segment, st.ex_rank, st.ex_world, st.ex_max_len = ex._addr, ex.rank, ex.world, ex.max_len
                st.ex_k, st.ex_timeout_s = __import__("ctypes").pointer(ex._kc), ex.timeout
                self._shard_view = np.ctypeslib.as_array(st.shard_sums)[:ex.world]
                self._plan_view = np.ctypeslib.as_array(st.plan_totals)[:ex.world]
            # the first moves of a due resample from C as well: the shard plan and, when the children stay with their
            # ancestors, this shard's weight-only prefix (ParticleShardGroup.resample finds both done)
            comm = self._comm
            st.plan_enabled = int(comm.placement == "local" and type(r) is LiuWestResampler
                                  and getattr(r, "_device_rng", False) and n <= r._segment_limit)
            if st.plan_enabled:
                st.plan_seed, st.plan_epoch = comm.seed & _U64, comm._epoch + 1
                st.plan_prefix_seed = (r._seed + 0x9E3779B97F4A7C15 * (comm.rank + 1)) & _U64
                st.plan_n_total, st.plan_tol = self.n_particles_global, float(comm.rebalance_tol)
        lw.adopt = 0
        lw.prefix = int(key is not None)
        self._step_arms = self._eng.STEP_ARMED if key is not None else None
        lw.enabled = 0
        if key is not None:
            lw.n_out, lw.seed, lw.epoch = key[1], key[2] & _U64, key[3]
            # the resample itself is queued from C only when nothing sits between this update and it: the stock
            # resampler class, a cloud that does not move between data, moments that came with the update (d <= 4),
            # same-size output (the spare buffer ping-pongs with the cloud), no divergence tracking
            fc = self._fused_canon
            if fc is not None and not self._eng.fused_canon_applies(d, n, key[1]):
                fc = None
            big_ok = d == 16 and self._eng.fused_canon_applies(d, n, key[1])       # (the split d = 16 sampler)
            if (type(r) is LiuWestResampler and self._timestep_identity and key[1] == n
                    and getattr(self.model, "_native_timestep", None) is None and (d <= 4 or big_ok)
                    and self._resampling_divergences is None):
                if self._x_spare is None or self._x_spare.shape != x.shape:
                    self._x_spare = self._eng.empty(d, n)
                lw.enabled = 1
                # ... and is taken as done without re-deriving it in Python (`_adopt_queued`): nothing but the stock
                # resampler's own arithmetic stands between the n_ess test and the new cloud, and C has done exactly that
                lw.adopt = int(not self._debug_resampling and not _NO_ADOPT)
                lw.postselect, lw.maxiter = int(bool(r._postselect)), int(r._maxiter)
                lw.a, lw.h, lw.zero_cov_comp = float(r._a), float(r._h), float(r._zero_cov_comp)
                lw.x_out, lw.ldx_out = self._x_spare.data_ptr(), self._x_spare.stride(0)
                lw.canon_kind = 0
                if fc is not None and d == 16:
                    lw.canon_kind, lw.canon_allow_sub = fc[0], int(bool(fc[2]))
                    lw.canon_basis = fc[1].data_ptr() if fc[1] is not None else None
        self._step_key = (r, self.resample_thresh)
        self._step_synced = True

    def _update_step(self, outcome, expparams, check_for_resample):
        """`update` for a native model on one cloud: qsmc_step does the fused update, the guards' tests, the commit,
        n_ess / min_n_ess and the resample test in C (and queues a due Liu-West resample itself); what is left here is
        the bookkeeping the reference keeps in Python objects (smc.py:388-457)."""
        self._data_record.append(outcome)
        self._just_resampled = False
        st = self._st
        if not self._step_synced or self._step_key[0] is not self.resampler or self._step_key[1] != self.resample_thresh:
            self._step_sync()
        elif not st.w_alt:                            # (the update after a reset / resample committed into implicit weights:
            st.w_alt = self._scratch_weights().data_ptr()      #  the struct, not the Python mirror, says a buffer is missing)
        fill = self._ep_fill
        if fill is not None and fill(self._ep, expparams):
            ep_ref = self._ep_ref
        else:
            exps = self.model._native_expparams(expparams)
            if len(exps) != 1:
                raise ValueError("update() takes exactly one experiment")
            ep_ref = exps[0]
        st.check_for_resample = check_for_resample
        eng = self._eng
        if self._st_exchange is None:
            eng.step(self._st_ref, self._desc_ref, ep_ref, outcome if type(outcome) is int else _as_int_outcome(outcome))
        else:
            try:
                eng.step(self._st_ref, self._desc_ref, ep_ref, _as_int_outcome(outcome))
            except _native.PeerTimeoutError:          # (qsmc_host_allreduce's own status: QSMC_ERR_TIMEOUT)
                raise RuntimeError("HostExchange: a peer did not arrive within {} s".format(
                    self._st_exchange.timeout)) from None
            self._shard_sums = self._shard_view.copy()       # every shard's sum w' (the next resample plan's input)
            if st.status & _native.STEP_PLAN_READY:
                self._step_plan = (st.plan_epoch, self._plan_view.copy(), bool(st.status & _native.STEP_PREFIX_QUEUED),
                                   st.update_token)
        eng.update_gen = st.update_token
        # what the call left armed on the handle: the gated prefix, if the struct carries a prefix key and the n_ess test
        # was asked for (qsmc_step disarms it otherwise)
        eng._armed_prefix = self._step_arms if check_for_resample else None
        status = st.status
        w_out = self._w_alt
        if status & _native.STEP_GUARD:
            # a guard is due: nothing was committed; the reference's own sequence, from the sums
            us = st.stats
            d = self._x.shape[0]
            mom = self._st_mom[:d + d * (d + 1) // 2].copy() if d <= 4 else None
            self._step_synced = False
            if self._moments_cache is _FROM_STEP:
                # the marker pointed at st.moments, which now hold the sums of this UNCOMMITTED update: should the policy
                # leave without committing ('skip', or 'error' caught by the caller) the moments are recomputed from the
                # unchanged weights, not read from there
                self._moments_cache = None
            return self._finish_update(us.sum, us.sumsq, us.min, us.n_bad, w_out, mom, expparams, check_for_resample)
        flush = getattr(self.resampler, "_flush_failed_warning", None)
        if flush is not None:
            flush()                       # the stream was just synchronised: deferred resampler warning
        # mirror the commit C made in the struct
        self._w, self._w_alt = w_out, self._w
        self._norm, self._sumsq = st.norm, st.sumsq
        self._moments_cache = _FROM_STEP if self._x.shape[0] <= 4 else None
        self._w_token = st.update_token
        self._view_version += 1
        self._normalization_record.append(st.stats.sum)
        self._min_n_ess = st.min_n_ess
        if not self._timestep_identity or getattr(self.model, "_native_timestep", None) is not None:
            self._timestep(expparams)
        if status & (_native.STEP_SMALL_ESS | _native.STEP_RESAMPLE_DUE):
            queued = bool(status & _native.STEP_RESAMPLE_QUEUED)
            if queued and st.lw.adopt:
                r, lw = self.resampler, st.lw
                # (the struct was filled at the last _step_sync: a resampler whose parameters were edited in place since --
                #  `upd.resampler.a = 0.9` -- must get ITS resample, so the queued one is taken only if it used them)
                if (r._a == lw.a and r._h == lw.h and r._zero_cov_comp == lw.zero_cov_comp and r._epoch + 1 == lw.epoch
                        and (r._seed & _U64) == lw.seed and int(bool(r._postselect)) == lw.postselect
                        and int(r._maxiter) == lw.maxiter and r._default_n_particles in (None, lw.n_out)):
                    return self._adopt_queued(status)
                queued = False                        # the resampler's own call runs it again, with its parameters
            if queued:
                # the square root the library formed for the resample it queued: sqrtm_psd of exactly this matrix, by the
                # routine the resampler is about to call -- handed over so that the host does not repeat it (40 us at
                # d = 16) while the GPU is already sampling; the resampler takes it only for a bit-identical covariance
                d = self._x.shape[0]
                cov_c = self._st_cov[:d * d].reshape(d, d)
                cov_bytes = cov_c.tobytes()                   # (before any substitute: what est_covariance_mtx will form)
                if not cov_c.any():
                    cov_c = st.lw.zero_cov_comp * np.eye(d)
                self._queued_sqrt = (cov_c.tobytes(), st.lw.h, self._st_S[:d * d].reshape(d, d).copy(), st.S_err)
                # the covariance's smallest eigenvalue came out of the same Jacobi: est_covariance_mtx's PSD check for
                # exactly this matrix needs no second eigendecomposition (distributions._cov_from_sums)
                self._queued_psd = (cov_bytes, st.cov_lambda_min)
            if queued and self._x.shape[0] > 4:
                # the moments pass ran inside the call (its result drove the queued resample): what eng.moments returns
                d = self._x.shape[0]
                mb = self._st_mom_big[:1 + d + d * (d + 1) // 2]
                self._moments_cache = (float(mb[0]), mb[1:1 + d].copy(), eng._unpack_upper(mb[1 + d:], d))
            self._maybe_resample(np.float64(st.n_ess), queued)

    def _adopt_queued(self, status):
        """The resample qsmc_step queued IS this update's resample (smc.py:263-277 -> 491-551 with
        resamplers.py:256-392 inside): mean, covariance, its zero-norm substitute, S = h sqrtm_psd(cov) and the sampler
        call were made in C with the resampler's own parameters, into the spare cloud.  What is left here is what the
        reference does around that arithmetic: its warnings (from the flags C left: n_ess, the covariance, its smallest
        eigenvalue), the counters, the swap, canonicalize if the kernels did not fold it in, clear_cache.  Round 4: the
        Python re-derivation this replaces (moments -> covariance -> eigvalsh -> the same call again, 120 us warm and
        ~400 us on the cold caches of the first resample after a reset) kept the GPU idle after a d = 16 resample's
        last kernel -- the "136 us after k_tomo_canon_list" of the round-3 traces."""
        st, r = self._st, self.resampler
        d, n = self._x.shape
        if status & _native.STEP_SMALL_ESS:                          # smc.py:267-271
            warnings.warn("Extremely small n_ess encountered ({}). Resampling is likely to fail. "
                          "Consider adding particles, or resampling more often.".format(np.float64(st.n_ess)),
                          ApproximationWarning)
        self._eng.step_adopted()                                     # (counted by the party that decides: qsmc_step_stats)
        self._just_resampled = True                                  # (update() cleared it: no 'without additional data')
        self._resample_count += 1
        # distributions.py:392-399 (via est_covariance_mtx): `not np.all(la.eig(cov)[0] >= 0)`.  The sign test is taken on
        # the smallest diagonal entry the Jacobi sweeps converged to; for an eigenvalue that is rounding noise around zero
        # (|lambda| <~ eps ||cov||: a singular covariance, e.g. tomography's x_0 = 1/2 row) LAPACK's `eig` and a Jacobi
        # iteration may land on different sides of zero, as two LAPACK builds may -- the warning is then a coin toss in the
        # reference too; no tolerance is applied, so that a genuinely indefinite covariance warns exactly as before.
        if st.cov_lambda_min < 0:
            warnings.warn('Numerical error in covariance estimation causing positive semidefinite '
                          'violation.', ApproximationWarning)
        if not self._st_cov[:d * d].any():                           # resamplers.py:283-290
            warnings.warn("Covariance has zero norm; adding in small covariance in resampler. "
                          "Consider increasing n_particles to improve covariance estimates.", ResamplerWarning)
        # the resampler's own state moves on as if it had been called (its Philox epoch keys the draw C made)
        r._epoch += 1
        assert r._epoch == st.lw.epoch
        r._pending_failed = self._eng                                # "failed to find valid models": at the next sync
        self._x, self._x_spare = self._x_spare, self._x              # the new cloud; the old one is the next spare
        self._w = self._w_alt = None                                 # uniform weights 1 / N, held implicitly
        self._norm = self._sumsq = float(n)
        self._invalidate()
        if self._canonicalize and not st.lw.canon_kind:              # smc.py:529 (the d = 16 kernels fold it in)
            self._canonicalize_device()
        try:
            self.model.clear_cache()
        except Exception as e:  # noqa: BLE001  (reference demotes these to warnings, smc.py:533-536)
            warnings.warn("Exception raised when clearing model cache: {}. Ignoring.".format(e))

    def update(self, outcome, expparams, check_for_resample=True):
        """One Bayes step (smc.py:388-457)."""
        if self._st is not None:
            return self._update_step(outcome, expparams, check_for_resample)
        self._data_record.append(outcome)
        self._just_resampled = False
        eng = self._eng
        w_out = self._scratch_weights()
        fused_moments = None
        if self._native:
            exps = self.model._native_expparams(expparams)
            if len(exps) != 1:
                raise ValueError("update() takes exactly one experiment")
            d = self._x.shape[0]
            if self._comm is not None:
                # sharded: this shard's sums land in pinned host memory like the single-GPU path, then ONE
                # small all-gather (shared memory on one host, else the backend's) makes them global
                n_mom = d + d * (d + 1) // 2 if d <= 4 else 0
                eng.arm_resample_prefix(None)        # (the n_ess test of a sharded cloud needs every shard's sums)
                if self._comm.device_transport:
                    # RCCL on the launch stream: the collective starts from the device vector the reducing kernel
                    # wrote -- no host round trip between the update and the all-reduce, one wait per datum
                    eng.update_fused(self._desc, self._x, self._w, w_out, self._norm, exps[0],
                                     _as_int_outcome(outcome), sync=False)
                    norm, sumsq, wmin, n_bad = self._comm.allreduce_update_stats_device(eng, 4 + n_mom)
                else:
                    st = eng.update_fused(self._desc, self._x, self._w, w_out, self._norm, exps[0],
                                          _as_int_outcome(outcome), moments="raw" if n_mom else False)
                    norm, sumsq, wmin, n_bad = self._comm.allreduce_update_stats(
                        eng, st.sum, st.sumsq, st.min, st.n_bad, eng._mom[d] if n_mom else None)
                self._shard_sums = self._comm.last_shard_sums
                if n_mom:
                    fused_moments = self._comm.last_extra            # (a copy: packed sums, see _moments)
                st = None
            elif d <= 4:
                eng.arm_resample_prefix(self._prefix_key() if check_for_resample else None)
                # the kernel also returns sum w'x, sum w'xx^T of the new weights (x is in registers
                # anyway): est_mean / est_covariance_mtx / the resampler need no further pass
                st = eng.update_fused(self._desc, self._x, self._w, w_out, self._norm, exps[0],
                                      _as_int_outcome(outcome), moments="raw")
                fused_moments = eng._mom[d].copy()                   # packed sums, unpacked lazily (_moments)
            else:
                eng.arm_resample_prefix(self._prefix_key() if check_for_resample else None)
                st = eng.update_fused(self._desc, self._x, self._w, w_out, self._norm, exps[0],
                                      _as_int_outcome(outcome))
        elif self._uk is not None:
            # a user model compiled into the fused update kernel: ONE pass over the cloud, like a native model
            eng.arm_resample_prefix(None)
            epm = _user_ep_matrix(expparams)
            if epm.shape[0] != 1:
                raise ValueError("update() takes exactly one experiment")
            d = self._x.shape[0]
            want_mom = d <= 4 and self._comm is None
            st = eng.update_user(self._uk, self._x, self._w, w_out, self._norm, epm[0], _as_int_outcome(outcome),
                                 moments=want_mom)
            if want_mom:
                fused_moments = eng._mom[d].copy()
            if hasattr(self.model, "count_likelihood_calls"):
                self.model.count_likelihood_calls(1, self._x.shape[1], 1)
        else:
            eng.arm_resample_prefix(None)             # (no fused update on this path: nothing may stay armed on the handle)
            L = self._device_likelihood(outcome, expparams)
            if L.shape[0] != 1 or L.shape[1] != 1:
                raise ValueError("update() takes exactly one outcome and one experiment")
            st = eng.update_from_likelihood(L.reshape(-1), self._weights(), w_out, self._norm)
        if st is not None:
            norm, sumsq, wmin, n_bad = self._reduce_stats(st)
        return self._finish_update(norm, sumsq, wmin, n_bad, w_out, fused_moments, expparams, check_for_resample)

    def _finish_update(self, norm, sumsq, wmin, n_bad, w_out, fused_moments, expparams, check_for_resample):
        """Everything of `update` after the sums are known (smc.py:369-457): guards, policies, commit, records,
        time step, n_ess, resample test."""
        eng = self._eng
        flush = getattr(self.resampler, "_flush_failed_warning", None)
        if flush is not None:
            flush()                       # the stream was just synchronised: deferred resampler warning
        fixed = 1.0 if abs(norm) < _EPS else norm                    # smc.py:369-370
        new_norm = fixed

        if n_bad > 0:                                                # smc.py:416-418
            warnings.warn("Negative weights occured in particle approximation. Smallest weight observed "
                          "== {}. Clipping weights.".format(wmin / fixed), ApproximationWarning)
            cst = eng.clip_weights(w_out, fixed)
            sum_w, sumsq, _, _ = self._reduce_stats(cst)
            new_norm = 1.0                                           # clipped weights are stored as-is
        else:
            sum_w = norm / fixed

        if sum_w <= self._zero_weight_thresh or not (sum_w == sum_w):   # smc.py:423-436
            pol = self._zero_weight_policy
            if pol == 'ignore':
                pass
            elif pol == 'skip':
                return
            elif pol == 'warn':
                warnings.warn("All particle weights are zero. This will very likely fail quite badly.",
                              ApproximationWarning)
            elif pol == 'error':
                raise RuntimeError("All particle weights are zero.")
            elif pol == 'reset':
                warnings.warn("All particle weights are zero. Resetting from initial prior.",
                              ApproximationWarning)
                self.reset(reset_weights=False)
            else:
                raise ValueError("Invalid zero-weight policy {} encountered.".format(pol))

        # commit: swap the weight buffers (smc.py:441)
        self._w, self._w_alt = w_out, self._w
        self._norm = float(new_norm)
        self._sumsq = float(sumsq)
        self._invalidate(locations=False)
        if self._native and n_bad == 0:
            # these weights are exactly what update number `update_gen` of the engine wrote: a resample that
            # follows may take its chunk sums from that kernel's tile sums (resamplers._arm_update_sums)
            self._w_token = eng.update_gen
        if fused_moments is not None and n_bad == 0 and new_norm != 0:
            self._moments_cache = ("packed", sum_w, fused_moments, new_norm)
        self._normalization_record.append(norm)                      # smc.py:444
        self._timestep(expparams)

        ess = self.n_ess                                             # smc.py:452-453
        if ess <= self._min_n_ess:
            self._min_n_ess = ess
        if check_for_resample:
            self._maybe_resample(ess)

    def _timestep(self, expparams):
        """Model.update_timestep between data (smc.py:447-449)."""
        step = getattr(self.model, "_native_timestep", None)
        if self._native and step is not None:
            # a random-walk model with device kernels: the cloud takes its step in place
            step(self, expparams)
            self._moments_cache = None
            self._host_locs = None
        elif not self._timestep_identity:
            dev_fn = getattr(self.model, "update_timestep_device", None)
            if dev_fn is not None and not self._native:
                # device-side plugin hook: the model moves the cloud where it lives
                self._x = self._check_cloud(dev_fn(self._x, expparams), "update_timestep_device")
            else:
                # plugin slow path: a model that moves particles between data and has no device step -- a user
                # model, or a decorator over one (DerivedModel forwards update_timestep)
                locs = self.model.update_timestep(self._host_locations(), expparams)[:, :, 0]
                self._x = self._eng.locs_to_soa(locs)
            self._invalidate()

    def batch_update(self, outcomes, expparams, resample_interval=5):
        """Update on a batch of data with the ESS test every `resample_interval` data
        (smc.py:459-487).

        The reference loops `update(..., check_for_resample=False)`.  Between two ESS tests nothing
        but a scalar renormalisation separates consecutive data, so for native models the data of
        one window are applied in ONE pass over the cloud (`qsmc_update_multi`, up to 8 per launch)
        and every per-datum record (normalization_record, min_n_ess, data_record) is rebuilt from the
        per-datum sums the kernel returns.  If any guard would have fired inside a window (negative
        weights, all-zero weights) the window is discarded -- the weights are double-buffered -- and
        replayed datum by datum, so warnings/exceptions/policies behave exactly as in the loop."""
        outcomes = np.asarray(outcomes)
        n_exps = outcomes.shape[0]
        if expparams.shape[0] != n_exps:
            raise ValueError("The number of outcomes and experiments must match.")
        if len(expparams.shape) == 1:
            expparams = expparams[:, None]
        fast = ((self._native or (self._uk is not None and self._timestep_identity)) and self._batch_fast_path
                and getattr(self.model, "_native_timestep", None) is None)   # moving particles: one datum at a time
        idx = 0
        kmax = self._eng.MULTI_KMAX
        while idx < n_exps:
            window_end = min(n_exps, (idx // resample_interval + 1) * resample_interval)
            stop = min(window_end, idx + kmax)
            # a single datum is served best by the dedicated one-datum kernel (16 B/lane loads)
            if not (fast and stop - idx >= 2 and self._fused_window(outcomes[idx:stop], expparams[idx:stop])):
                for j in range(idx, stop):
                    self.update(outcomes[j], expparams[j], check_for_resample=False)
            idx = stop
            if idx % resample_interval == 0:
                self._maybe_resample()

    _batch_fast_path = True

    def _fused_window(self, outcomes, expparams):
        """Apply len(outcomes) <= 8 data in one kernel pass.  Returns False (state untouched) if a
        guard of the per-datum loop would have fired, so the caller can replay it faithfully."""
        eng = self._eng
        k = len(outcomes)
        w_out = self._scratch_weights()
        if self._uk is not None:
            # a compiled user model (likelihood_hip): the window through its own JIT window kernel
            epm = np.vstack([_user_ep_matrix(expparams[j]) for j in range(k)])
            if epm.shape[0] != k:
                return False
            outs = [_as_int_outcome(outcomes[j]) for j in range(k)]
            stats, m1, m2 = eng.update_multi_user(self._uk, self._x, self._w, w_out, self._norm, epm, outs)
            if hasattr(self.model, "count_likelihood_calls"):
                self.model.count_likelihood_calls(1, self._x.shape[1], k)
        else:
            if self._x.shape[0] > _native.QSMC_MAX_D:
                # wide clouds (16 < d <= 64): the window kernel takes sparse measurement vectors only (at most four rows
                # per datum: qsmc_update_multi); anything else goes datum by datum
                meas = np.asarray(expparams['meas']).reshape(k, -1)
                if meas.shape[1] != self._x.shape[0] or np.count_nonzero(meas, axis=1).max() > 4 \
                        or np.count_nonzero(meas, axis=1).min() < 1 or getattr(self.model, "_pow", None) is not None:
                    return False
            exps, outs = [], []
            for j in range(k):
                e = self.model._native_expparams(expparams[j])
                if len(e) != 1:
                    return False
                exps.append(e[0])
                outs.append(_as_int_outcome(outcomes[j]))
            stats, m1, m2 = eng.update_multi(self._desc, self._x, self._w, w_out, self._norm, exps, outs)
        if self._comm is not None:
            # sharded: the window's per-datum sums (and the moment sums of its last datum) are additive over the
            # shards -- one small reduction for the whole window instead of one per datum
            d = self._x.shape[0]
            vec = np.empty(3 * k + 1 + (0 if m1 is None else d + d * d))
            for j, st in enumerate(stats):
                vec[3 * j:3 * j + 3] = st.sum, st.sumsq, st.n_bad
            vec[3 * k] = min(st.min for st in stats)
            if m1 is not None:
                vec[3 * k + 1:3 * k + 1 + d] = m1
                vec[3 * k + 1 + d:] = m2.reshape(-1)
            tot, rows = self._comm.allreduce_host_vector(vec, min_index=3 * k)

            class _St:
                __slots__ = ("sum", "sumsq", "n_bad", "min")
            red = []
            for j in range(k):
                st = _St()
                st.sum, st.sumsq, st.n_bad, st.min = tot[3 * j], tot[3 * j + 1], tot[3 * j + 2], tot[3 * k]
                red.append(st)
            stats = red
            if m1 is not None:
                m1, m2 = tot[3 * k + 1:3 * k + 1 + d].copy(), tot[3 * k + 1 + d:].reshape(d, d).copy()
            shard_sums = rows[:, 3 * (k - 1)].copy()
        flush = getattr(self.resampler, "_flush_failed_warning", None)
        if flush is not None:
            flush()
        prev = None
        norms = []
        for st in stats:
            nk = st.sum if prev is None else (st.sum / prev if prev != 0 else np.nan)
            if st.n_bad > 0 or not (abs(nk) >= _EPS):      # negative / NaN weights, or the zero-weight test
                return False
            norms.append(nk)
            prev = st.sum
        # commit the window
        for j in range(k):
            self._data_record.append(outcomes[j])
            self._normalization_record.append(norms[j])
            ess = np.float64(stats[j].sum) * np.float64(stats[j].sum) / np.float64(stats[j].sumsq)
            if ess <= self._min_n_ess:
                self._min_n_ess = ess
        self._just_resampled = False
        self._w, self._w_alt = w_out, self._w
        self._norm = float(stats[-1].sum)
        self._sumsq = float(stats[-1].sumsq)
        self._invalidate(locations=False)
        # (these weights are what update number `update_gen` -- the window's pass -- wrote: a resample that follows takes its
        #  chunk sums from that kernel's tile sums, resamplers._arm_update_sums)
        self._w_token = eng.update_gen
        if self._comm is not None:
            self._shard_sums = shard_sums
        if m1 is not None:
            self._moments_cache = (1.0, m1 / self._norm, m2 / self._norm)
        return True

    # ------------------------------------------------------------------ experiment design
    def _hyp_sums(self, expparams, what=3):
        """Per experiment: all outcomes' hypothetical sums (qsmc_hypothetical_sums_multi: one call for the whole design;
        `what`: the columns the caller reads, Engine.HYP_LOG | Engine.HYP_MOMENTS)."""
        eng = self._eng
        shift = self.est_mean()
        model = self.model
        # the first experiment is translated and queued on its own: the GPU works on its passes while the host translates
        # the rest of the design (records -> C structs, outcome domains: ~10 us per experiment), instead of idling until
        # the whole design has been prepared (qsmc_hypothetical_sums_begin / _collect, round 5)
        if expparams.shape[0] == 1:
            # one experiment: nothing to overlap the translation of a "rest" with -- one call, one wait
            sums = eng.hypothetical_sums_multi(self._desc, self._x, self._w, self._norm, model._native_expparams(expparams),
                                               [dom.values for dom in model.domain(expparams)], shift, what)
            return self._reduce_design_sums(sums)
        head = expparams[:1]
        jobs = [eng.hypothetical_sums_begin(self._desc, self._x, self._w, self._norm, model._native_expparams(head),
                                            [dom.values for dom in model.domain(head)], shift, what)]
        try:
            if expparams.shape[0] > 1:
                rest = expparams[1:]
                jobs.append(eng.hypothetical_sums_begin(self._desc, self._x, self._w, self._norm, model._native_expparams(rest),
                                                        [dom.values for dom in model.domain(rest)], shift, what))
        finally:
            eng.hypothetical_sums_collect()          # (also after an error in the second half: nothing stays in flight)
        return self._reduce_design_sums([r for job in jobs for r in job.rows])

    def _reduce_design_sums(self, sums):
        if self._comm is not None:
            # every entry is a sum over particles with the GLOBAL normaliser and a shift all ranks agree on (the global
            # mean): additive over the shards (columns nobody asked for are NaN on every shard); one reduction per design
            flat = np.concatenate([a.reshape(-1) for a in sums])
            tot = self._comm.allreduce_host_vector(flat)[0]
            out, at = [], 0
            for a in sums:
                out.append(tot[at:at + a.size].reshape(a.shape))
                at += a.size
            sums = out
        return sums

    def bayes_risk(self, expparams):
        """Bayes risk (quadratic loss, scale matrix Q) of hypothetical experiments: the expected
        posterior variance  sum_o N[o] var[o]  for each experiment (smc.py:553-611).

        Native models: one fused pass per experiment yields, for every outcome, the hypothetical
        normalisation and the (mean-shifted) first and second moments; nothing of size
        n_outcomes x N is materialised.  Other models go through `hypothetical_update`."""
        expparams = np.atleast_1d(expparams).reshape(-1)
        if self._native and self._eng.hyp_row_width(self._desc, self._x.shape[0]) > 2:      # (the kernels carry the moments)
            d = self._x.shape[0]
            Q = np.asarray(self.model.Q, dtype=np.float64)
            risk = np.empty(expparams.shape[0])
            sums = self._hyp_sums(expparams, self._eng.HYP_MOMENTS)
            if len({a.shape for a in sums}) == 1:
                sums = [np.stack(sums)]                      # equal outcome counts: the whole design in one set of array operations
            at = 0
            for a in sums:
                a = a.reshape((-1,) + a.shape[-2:])
                N, s1, s2 = a[..., 0:1], a[..., 2:2 + d], a[..., 2 + d:2 + 2 * d]
                with np.errstate(divide='ignore', invalid='ignore'):
                    var = np.where(N > 0, s2 - s1 * s1 / N, 0.0)
                # (Q is the vector of smc.py:598's `Q * (x - mu) ** 2`; a square Q, which the per-experiment form of this
                #  method also took, contracts to one more axis: everything after the experiment axis is summed)
                risk[at:at + a.shape[0]] = (var @ Q).reshape(a.shape[0], -1).sum(axis=-1)
                at += a.shape[0]
            return risk
        return self._design_generic(expparams, "risk")

    def expected_information_gain(self, expparams):
        """Expected KL divergence posterior||prior over the outcomes of each hypothetical
        experiment,  sum_o N[o] KLD[o]  (smc.py:613-663).  `0 log 0` is taken as 0."""
        expparams = np.atleast_1d(expparams).reshape(-1)
        if self._native and self._x.shape[0] <= _native.QSMC_MAX_D:      # (d > 16: through qsmc_likelihood, below)
            eig = np.empty(expparams.shape[0])
            sums = self._hyp_sums(expparams, self._eng.HYP_LOG)
            if len({a.shape for a in sums}) == 1:
                sums = [np.stack(sums)]
            at = 0
            for a in sums:
                a = a.reshape((-1,) + a.shape[-2:])
                N, sl = a[..., 0], a[..., 1]
                with np.errstate(divide='ignore', invalid='ignore'):
                    eig[at:at + a.shape[0]] = np.sum(np.where(N > 0, sl - N * np.log(N), 0.0), axis=-1)
                at += a.shape[0]
            return eig
        return self._design_generic(expparams, "eig")

    def _design_generic(self, expparams, what):
        """Any model whose kernels carry no design sums (plugin models three ways, tomography's risk, wide tomography):
        the same one-pass sums the native design kernels form -- per (outcome, experiment) N = sum w L, sum w L ln L,
        sum w L (x - c), sum w L (x - c)^2 with c the current mean -- as torch products on the device from the model's
        likelihood tensor; a sharded updater adds the shards' sums (they are sums over particles with the global
        normaliser and a shift every rank agrees on), then the reference's formulas (smc.py:586-611, 640-663) on the totals."""
        n_eps = expparams.shape[0]
        if n_eps > 1 and not self.model.is_n_outcomes_constant:
            return np.array([self._design_generic(expparams[i:i + 1], what)[0] for i in range(n_eps)])
        Q = np.asarray(self.model.Q, dtype=np.float64)
        if what == "risk" and Q.ndim != 1:
            return self._design_generic_host(expparams, what)
        t = self._eng.torch
        os_ = self.model.domain(expparams[0:1])[0].values
        n_o, (d, n) = len(os_), self._x.shape
        w = self._eng.normalized_weights(self._weights(), self._norm)
        per = 1 + (2 * d if what == "risk" else 1)
        rows = np.empty((n_o, n_eps, per))
        if what == "risk":
            xc = self._x - self._eng.to_device(np.asarray(self.est_mean(), dtype=np.float64))[:, None]
            xc2 = xc * xc
        group = max(1, int(1e8 // max(1, n_o * n))) if n_eps > 1 else 1        # (the likelihood tensor: <= ~1e8 doubles at a time)
        for e0 in range(0, n_eps, group):
            L = self._device_likelihood(os_, expparams[e0:e0 + group])          # (n_o, g, N) device
            g = L.shape[1]
            wl = L * w
            cols = [wl.sum(dim=2, keepdim=True)]
            if what == "risk":
                flat = wl.reshape(n_o * g, n)
                cols += [(flat @ xc.T).reshape(n_o, g, d), (flat @ xc2.T).reshape(n_o, g, d)]
            else:
                cols.append((wl * t.where(L > 0, t.log(L), t.zeros_like(L))).sum(dim=2, keepdim=True))
            rows[:, e0:e0 + g] = t.cat(cols, dim=2).cpu().numpy()
        if self._comm is not None:
            rows = self._comm.allreduce_host_vector(rows.reshape(-1))[0].reshape(rows.shape)
        N = rows[..., 0]
        with np.errstate(divide='ignore', invalid='ignore'):
            if what == "risk":
                s1, s2 = rows[..., 1:1 + d], rows[..., 1 + d:]
                var = np.where(N[..., None] > 0, s2 - s1 * s1 / N[..., None], 0.0)      # N var, per coordinate
                return (var @ Q).sum(axis=0)
            return np.sum(np.where(N > 0, rows[..., 1] - N * np.log(N), 0.0), axis=0)

    def _design_generic_host(self, expparams, what):
        """The reference's formulas on `hypothetical_update` output (a square scale matrix Q: one cloud only)."""
        self._single_cloud_only("bayes_risk with a matrix-valued Q")
        os_ = self.model.domain(expparams[0:1])[0].values
        w_hyp, N = self.hypothetical_update(os_, expparams, return_normalization=True)
        N = N[:, :, 0]
        locs = self.particle_locations
        mu = np.dot(w_hyp, locs)
        var = np.sum(w_hyp * np.sum(self.model.Q * (locs[None, None, :, :] - mu[:, :, None, :]) ** 2, axis=3), axis=2)
        return np.sum(N * var, axis=0)

    def risk(self, x0):
        return self.bayes_risk(np.array([(x0,)], dtype=self.model.expparams_dtype))

    # ------------------------------------------------------------------ resampling
    def _prefix_key(self):
        """What the resample after THIS update would ask the device for first (resamplers._prepare_device), so the
        update can queue it behind itself, gated on the device-side copy of the test in `_maybe_resample`
        (qsmc_lw_arm_prefix); None when that resample would not take the device-RNG path.  (On the per-datum path:
        attribute reads only.)"""
        r = self.resampler
        if not self._native or not getattr(r, "_device_rng", False):
            return None
        n = self._x.shape[1]
        if n > r._segment_limit:
            return None
        n_out = r._default_n_particles
        return (self.n_particles_global * self.resample_thresh, n if n_out is None else int(n_out), r._seed,
                r._epoch + 1)

    def _maybe_resample(self, ess=None, queued=False):
        ess = self.n_ess if ess is None else ess
        if ess <= 10:
            warnings.warn("Extremely small n_ess encountered ({}). Resampling is likely to fail. "
                          "Consider adding particles, or resampling more often.".format(ess),
                          ApproximationWarning)
        if ess < self.n_particles_global * self.resample_thresh:
            prepare = getattr(self.resampler, "_prepare_device", None)
            armed = self._eng._armed_prefix
            if (prepare is not None and self._comm is None and not queued
                    and not (armed is not None and self._w_token == self._eng.update_gen
                             and (armed is self._eng.STEP_ARMED or armed == self._prefix_key()))):
                # (with the prefix armed FOR THIS UPDATER'S latest fused update, the update queued it behind itself on
                #  the device's own ESS test: the resampler's call finds it done, or redoes it should the device have
                #  decided otherwise; a resample qsmc_step queued itself is further along still)
                prepare(self.model, self)            # GPU starts on the weight-only prefix right away
            self.resample(_defer_warning=True)

    def resample(self, _defer_warning=False):
        """Force a resampling step now (smc.py:491-551).

        When triggered from `update` with the device RNG, the step is fully asynchronous: nothing
        is read back, so the host queues the next update while the GPU resamples; the (rare)
        'failed to find valid models' warning is then issued at the next update's synchronisation.
        A direct call waits and warns immediately, like the reference."""
        if self.just_resampled:
            warnings.warn("Resampling without additional data; this may not perform as desired.",
                          ResamplerWarning)
        self._just_resampled = True
        self._resample_count += 1
        if self._debug_resampling:
            old_mean, old_cov = self.est_mean(), self.est_covariance_mtx()
        if self._resampling_divergences is not None:
            # (smc.py:506-510 copies the cloud; here the old device buffers simply stay referenced: the resampler
            #  writes a fresh cloud and the weight buffers are not reused before the next update)
            old_cloud = (self._x, self._w, self._norm)

        if self._comm is not None:
            new = self._comm.resample(self, self.resampler)
        else:
            can_defer = _defer_warning and hasattr(self.resampler, "_flush_failed_warning")
            if can_defer:
                self.resampler._defer_failed_check = True
            try:
                new = self.resampler(self.model, self)
            finally:
                if can_defer:
                    self.resampler._defer_failed_check = False
            if not _defer_warning and hasattr(self.resampler, "_flush_failed_warning"):
                self.resampler._flush_failed_warning(synchronize=True)
        if isinstance(new, ParticleDistribution):
            if new._x is self._x_spare:               # the resampler filled the spare cloud: the old one is the next spare
                self._x_spare = self._x if self._x.shape == new._x.shape else None
            self._x, self._w, self._norm, self._sumsq = new._x, new._w, new._norm, new._sumsq
            if self._comm is not None:           # uniform weights: a shard's weight total is its size
                self._shard_sums = np.asarray(self._comm.last_shard_sizes, dtype=np.float64)
        else:                                           # foreign resampler returning host arrays
            self._set_host(new.particle_locations, new.particle_weights)
        self._w_alt = None
        self._invalidate()
        if self._st is not None:
            self._st.lw.redraw_pending = 1            # (its failed-first-try count comes back with the next update's sums)
        if self._canonicalize and not getattr(new, "_canonicalized", False):
            self._canonicalize_device()
        try:
            self.model.clear_cache()
        except Exception as e:  # noqa: BLE001  (reference demotes these to warnings, smc.py:533-536)
            warnings.warn("Exception raised when clearing model cache: {}. Ignoring.".format(e))
        if self._resampling_divergences is not None:                  # smc.py:538-542
            self._resampling_divergences.append(self._kl_from_device(*old_cloud))
        if self._debug_resampling:
            import logging
            new_mean, new_cov = self.est_mean(), self.est_covariance_mtx()
            logging.getLogger(__name__).debug("Resampling changed mean by {}. Norm change in cov: {}.".format(
                old_mean - new_mean, np.linalg.norm(new_cov - old_cov)))
