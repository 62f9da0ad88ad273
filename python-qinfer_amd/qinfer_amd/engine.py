"""Device engine: owns the libqsmc_hip handle for one GPU and wraps each C-ABI entry point so
that it takes PyTorch-ROCm tensors (used ONLY as device-memory owners + stream providers).

Layout contract: particle locations are SoA `x[d, N]` float64 contiguous,
weights `w[N]` float64, kept unnormalised with a host-side normaliser.
"""
import ctypes as C
import threading

import numpy as np

from . import _native
from ._exceptions import NativeLibraryError

_engines = {}
_lock = threading.Lock()


def get_engine(device=None):
    """Engine for `device` (int index, torch.device or None = current)."""
    import torch
    if not torch.cuda.is_available():
        raise NativeLibraryError(
            "qinfer_amd needs an AMD GPU (torch.cuda.is_available() is False). The SMC path runs "
            "only on the HIP kernels in libqsmc_hip.so; there is no CPU fallback.")
    if device is None:
        idx = torch.cuda.current_device()
    elif isinstance(device, int):
        idx = device
    else:
        dev = torch.device(device)
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        if idx not in _engines:
            _engines[idx] = Engine(idx)
        return _engines[idx]


class Engine:
    def __init__(self, index):
        import torch
        self.torch = torch
        self.lib = _native.load()
        self.index = index
        self.device = torch.device("cuda", index)
        h = C.c_void_p()
        _native.check(None, self.lib.qsmc_create(C.byref(h), index), "qsmc_create")
        self.h = h
        self.update_gen = 0          # number of qsmc_update_fused calls on this handle (see use_update_sums)
        self._stats = torch.empty(4 + 4 + 10, dtype=torch.float64, device=self.device)   # stats + moments (d <= 4)
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._raw_stream = raw if raw is not None else (
            lambda idx: torch.cuda.current_stream(torch.device("cuda", idx)).cuda_stream)
        self._st = _native.UpdateStats()                 # reused across update calls
        self._mom = {d: np.empty(d + d * (d + 1) // 2, dtype=np.float64) for d in range(1, 5)}
        self._mom_ptr = {d: _native.f64_ptr(a) for d, a in self._mom.items()}
        self._st_ref = C.byref(self._st)
        self._armed_prefix = None    # what qsmc_lw_arm_prefix was last told (arm_resample_prefix)
        self._qsmc_step = self.lib.qsmc_step
        self._h_int = h.value        # (the handle as a plain int: marshalled fastest on the per-datum call)
        self._design_jobs = []       # design passes begun and not yet collected (hypothetical_sums_begin / _collect)

    def _no_design_in_flight(self, what):
        """Between `hypothetical_sums_begin` and `_collect` the pinned block belongs to the queued design passes: nothing
        else that reduces through it may be asked of the engine."""
        if self._design_jobs:
            raise RuntimeError("{} while design passes are in flight: call hypothetical_sums_collect() first".format(what))

    # ------------------------------------------------------------------ memory / streams
    def stream(self):
        # raw hipStream_t of torch's current stream (a plain int; ~0.3 us vs ~4.5 us for the Stream object)
        return self._raw_stream(self.index)

    def empty(self, *shape, dtype=None):
        return self.torch.empty(*shape, dtype=dtype or self.torch.float64, device=self.device)

    def to_device(self, arr, dtype=None):
        t = self.torch.from_numpy(np.ascontiguousarray(arr))
        if dtype is not None:
            t = t.to(dtype)
        return t.to(self.device)

    def locs_to_soa(self, locs):
        """(N, d) host AoS -> (d, N) device SoA."""
        locs = np.asarray(locs, dtype=np.float64)
        return self.to_device(np.ascontiguousarray(locs.T))

    @staticmethod
    def _p(t):
        return t.data_ptr()            # argtypes are c_void_p: a plain int is marshalled fastest

    _triu_cache = {}

    @classmethod
    def _unpack_upper(cls, packed, d):
        """Row-major upper triangle (d(d+1)/2) -> symmetric (d, d)."""
        if d == 1:
            return np.array([[packed[0]]])
        idx = cls._triu_cache.get(d)
        if idx is None:
            idx = cls._triu_cache[d] = np.triu_indices(d)
        s2 = np.zeros((d, d))
        s2[idx] = packed
        return s2 + np.triu(s2, 1).T

    def _chk(self, rc, what):
        _native.check(self.h, rc, what)

    def device_cus(self):
        """(usable, reported) compute units: the census the library sizes its grid-barrier kernels by."""
        a, b = C.c_int32(), C.c_int32()
        self._chk(self.lib.qsmc_device_cus(self.h, C.byref(a), C.byref(b)), "qsmc_device_cus")
        return a.value, b.value

    # ------------------------------------------------------------------ profiling hooks (bench.py)
    def set_profiling(self, enabled):
        """False/0: off; True/1: time every update / sampler launch; N > 1: every N-th launch of each kind."""
        self._chk(self.lib.qsmc_set_profiling(self.h, int(enabled)), "qsmc_set_profiling")

    def set_profiling_tags(self, tags=None):
        """Which kernel kinds carry events while profiling is on: an iterable of tag numbers, or None for all."""
        mask = 0 if tags is None else sum(1 << int(t) for t in tags)
        self._chk(self.lib.qsmc_set_profiling_tags(self.h, mask), "qsmc_set_profiling_tags")

    def profile_read(self, cap=4096):
        """(durations in ms, tags) of the timed kernels launched since profiling was enabled / last read,
        oldest first.  tag 0 = update kernel (explicit weights), 2 = update kernel (implicit weights),
        1 = the resampler's sampling kernel."""
        buf = (C.c_float * cap)()
        tags = (C.c_int32 * cap)()
        n = C.c_int32()
        self._chk(self.lib.qsmc_profile_read(self.h, buf, tags, cap, C.byref(n)), "qsmc_profile_read")
        return np.array(buf[:n.value], dtype=np.float64), np.array(tags[:n.value], dtype=np.int64)

    def last_update_kernel_ms(self):
        ms = C.c_float()
        self._chk(self.lib.qsmc_last_update_kernel_ms(self.h, C.byref(ms)), "qsmc_last_update_kernel_ms")
        return ms.value

    # ------------------------------------------------------------------ likelihood / validity
    def likelihood(self, desc, x, exps, outcomes):
        """L[n_o, n_e, N] on device for SoA x; `exps` list of ExpParam, `outcomes` int sequence."""
        n = x.shape[1]
        n_e, n_o = len(exps), len(outcomes)
        out = self.empty(n_o, n_e, n)
        if n == 0 or n_e == 0 or n_o == 0:
            return out
        ep = (_native.ExpParam * n_e)(*exps)
        oc = (C.c_int64 * n_o)(*[int(o) for o in outcomes])
        self._chk(self.lib.qsmc_likelihood(self.h, C.byref(desc), self._p(x), x.stride(0), n, ep, n_e,
                                           oc, n_o, self._p(out), self.stream()), "qsmc_likelihood")
        return out

    def are_models_valid(self, desc, x):
        n = x.shape[1]
        out = self.empty(n, dtype=self.torch.uint8)
        if n:
            self._chk(self.lib.qsmc_are_models_valid(self.h, C.byref(desc), self._p(x), x.stride(0), n,
                                                     self._p(out), self.stream()), "qsmc_are_models_valid")
        return out

    # ------------------------------------------------------------------ weight passes
    def update_fused(self, desc, x, w_in, w_out, prev_norm, exp, outcome, sync=True, moments=False):
        """Returns UpdateStats (or (UpdateStats, s1, s2) with moments=True, d <= 4: UNNORMALISED
        sum w' x and sum w' x x^T of the new weights, produced by the same kernel).  The returned
        UpdateStats object is reused by the next call: read it before updating again."""
        if self._design_jobs:
            self._no_design_in_flight("update_fused")
        d = x.shape[0]
        self.update_gen += 1                        # mirrors the handle's generation counter (qsmc_update_token)
        self._chk(self.lib.qsmc_update_fused(
            self.h, desc, x.data_ptr(), x.stride(0), x.shape[1],
            w_in.data_ptr() if w_in is not None else None, w_out.data_ptr(),
            prev_norm, exp, outcome, self._stats.data_ptr(),
            self._st_ref if sync else None, self._mom_ptr[d] if moments else None, self.stream()),
            "qsmc_update_fused")
        if not sync:
            return None
        if not moments or moments == "raw":         # "raw": the packed sums stay in self._mom[d] for the caller
            return self._st
        mom = self._mom[d]
        return self._st, mom[:d].copy(), self._unpack_upper(mom[d:], d)

    STEP_ARMED = "armed by qsmc_step"      # value of _armed_prefix while the per-datum C path arms the prefix itself

    def step(self, st_ref, desc_ref, ep_ref, outcome):
        """One datum through qsmc_step (see include/qsmc.h): fused update + the no-guard tail of SMCUpdater.update
        + (when allowed and due) the Liu-West resample queued in C.  Results are in the qsmc_step_t behind st_ref."""
        if self._design_jobs:
            self._no_design_in_flight("step")
        rc = self._qsmc_step(self._h_int, st_ref, desc_ref, ep_ref, outcome, self._raw_stream(self.index))
        if rc:
            self._chk(rc, "qsmc_step")
        # (whether the call armed the gated prefix is the caller's to record in `_armed_prefix`: it filled the struct)

    def step_adopted(self):
        """The caller has taken the resample queued by the latest `step` as its own (counted by qsmc_step_stats)."""
        self._chk(self.lib.qsmc_step_adopted(self.h), "qsmc_step_adopted")

    def step_stats(self):
        """(resamples queued by qsmc_step, resamples whose caller-side call adopted the queued one)."""
        q, a = C.c_int64(), C.c_int64()
        self._chk(self.lib.qsmc_step_stats(self.h, C.byref(q), C.byref(a)), "qsmc_step_stats")
        return q.value, a.value

    MULTI_KMAX = 8

    def update_multi(self, desc, x, w_in, w_out, prev_norm, exps, outcomes):
        """K <= 8 data in one pass.  Returns (list of UpdateStats per datum, s1, s2) -- s1/s2 are the
        unnormalised moment sums of the final weights (None for d > 4)."""
        if self._design_jobs:
            self._no_design_in_flight("update_multi")
        k = len(exps)
        d = x.shape[0]
        ep = (_native.ExpParam * k)(*exps)
        oc = (C.c_int64 * k)(*[int(o) for o in outcomes])
        st = (_native.UpdateStats * k)()
        mom = np.empty(d + d * (d + 1) // 2, dtype=np.float64) if d <= 4 else None
        self._chk(self.lib.qsmc_update_multi(
            self.h, C.byref(desc), self._p(x), x.stride(0), x.shape[1],
            self._p(w_in) if w_in is not None else None, self._p(w_out), float(prev_norm), ep, oc, k, st,
            _native.f64_ptr(mom) if mom is not None else None, self.stream()), "qsmc_update_multi")
        self.update_gen += 1                        # (the window left tile sums like a fused update: qsmc_update_token)
        if mom is None:
            return list(st), None, None
        return list(st), mom[:d].copy(), self._unpack_upper(mom[d:], d)

    HYP_LOG, HYP_MOMENTS = 1, 2                     # qsmc.h: QSMC_HYP_LOG / QSMC_HYP_MOMENTS

    @staticmethod
    def hyp_row_width(desc, d):
        """Columns of a design row (include/qsmc.h: qsmc_hypothetical_sums_multi): [N, sum w L ln L] and, for the models whose
        kernels carry them (d <= 4, not tomography: its kernels are built for the maximal dimension), 2 d moment sums."""
        return 2 + 2 * d if (d <= 4 and desc.kind != _native.MODEL_TOMOGRAPHY) else 2

    def hypothetical_sums(self, desc, x, w, norm, exp, outcomes, shift, what=3):
        """(n_o, 2 + 2d) array [N, sum wL log L, sum wL (x-c), sum wL (x-c)^2] (d <= 4; else (n_o, 2)) of one experiment.
        `what`: the columns the caller reads (HYP_LOG: [1], HYP_MOMENTS: [2:]); binomial experiments leave the others NaN."""
        return self.hypothetical_sums_multi(desc, x, w, norm, [exp], [outcomes], shift, what)[0]

    def hypothetical_sums_multi(self, desc, x, w, norm, exps, outcomes, shift, what=3):
        """The same for several experiments in one call (qsmc_hypothetical_sums_multi: binomial experiments' passes queue
        back to back, one wait): `exps` ExpParam records, `outcomes` one outcome list per experiment; a list of arrays."""
        d = x.shape[0]
        per = self.hyp_row_width(desc, d)
        n_e = len(exps)
        counts = [len(o) for o in outcomes]
        total = sum(counts)
        out = np.empty((total, per), dtype=np.float64)
        oc = np.ascontiguousarray(np.concatenate([np.asarray(o).ravel() for o in outcomes]), dtype=np.int64)
        no = np.asarray(counts, dtype=np.int32)
        ep = (_native.ExpParam * n_e)(*exps)
        shift = np.ascontiguousarray(shift, dtype=np.float64)
        self._chk(self.lib.qsmc_hypothetical_sums_multi(
            self.h, C.byref(desc), self._p(x), x.stride(0), x.shape[1],
            self._p(w) if w is not None else None, float(norm), ep, n_e,
            oc.ctypes.data_as(C.POINTER(C.c_int64)), no.ctypes.data_as(C.POINTER(C.c_int32)),
            _native.f64_ptr(shift), int(what), _native.f64_ptr(out), self.stream()), "qsmc_hypothetical_sums_multi")
        res, at = [], 0
        for c in counts:
            res.append(out[at:at + c])
            at += c
        return res

    def hypothetical_sums_begin(self, desc, x, w, norm, exps, outcomes, shift, what=3):
        """Queue the design passes of these experiments and return without waiting (qsmc_hypothetical_sums_begin); the
        returned job's `rows` -- one array per experiment -- are valid after `hypothetical_sums_collect`.  Several jobs may be
        begun one after the other (a caller that prepares its experiments as it goes); nothing else may be asked of the
        engine in between."""
        d = x.shape[0]
        per = self.hyp_row_width(desc, d)
        n_e = len(exps)
        counts = [len(o) for o in outcomes]
        out = np.empty((sum(counts), per), dtype=np.float64)
        oc = np.ascontiguousarray(np.concatenate([np.asarray(o).ravel() for o in outcomes]), dtype=np.int64)
        no = np.asarray(counts, dtype=np.int32)
        ep = (_native.ExpParam * n_e)(*exps)
        shift = np.ascontiguousarray(shift, dtype=np.float64)
        self._chk(self.lib.qsmc_hypothetical_sums_begin(
            self.h, C.byref(desc), self._p(x), x.stride(0), x.shape[1],
            self._p(w) if w is not None else None, float(norm), ep, n_e,
            oc.ctypes.data_as(C.POINTER(C.c_int64)), no.ctypes.data_as(C.POINTER(C.c_int32)),
            _native.f64_ptr(shift), int(what), _native.f64_ptr(out), self.stream()), "qsmc_hypothetical_sums_begin")
        rows, at = [], 0
        for c in counts:
            rows.append(out[at:at + c])
            at += c
        job = type("DesignJob", (), {})()
        job.rows, job._keep = rows, (out, oc, no, ep, shift)          # (the C side holds pointers into these until collect)
        self._design_jobs = self._design_jobs + [job]
        return job

    def hypothetical_sums_collect(self):
        """Wait for every design pass queued by `hypothetical_sums_begin` and fill the jobs' rows."""
        try:
            self._chk(self.lib.qsmc_hypothetical_sums_collect(self.h, self.stream()), "qsmc_hypothetical_sums_collect")
        finally:
            self._design_jobs = []           # (also after an error: the library has dropped its queue -- chain2_flush)

    # ------------------------------------------------------------------ user models compiled at run time
    def user_kernel(self, source, d, n_ep):
        """Compile a model's `likelihood_hip` source into the fused update kernel (qsmc_user_kernel_build; cached per
        source).  Returns an object with `.ptr`, `.d`, `.n_ep`, `.has_valid`."""
        key = (source, int(d), int(n_ep))
        cache = self.__dict__.setdefault("_user_kernels", {})
        uk = cache.get(key)
        if uk is not None:
            return uk
        import os
        hint = os.path.join(os.path.dirname(self.torch.__file__), "lib", "libhiprtc.so")
        log = C.create_string_buffer(1 << 16)
        ptr = C.c_void_p()
        rc = self.lib.qsmc_user_kernel_build(self.h, source.encode(), int(d), int(n_ep),
                                             hint.encode() if os.path.exists(hint) else None, C.byref(ptr), log, len(log))
        if rc:
            msg = self.lib.qsmc_strerror(rc).decode()
            raise RuntimeError("likelihood_hip: qsmc_user_kernel_build failed ({}; {})\n{}".format(
                msg, self.lib.qsmc_last_hip_error(self.h).decode(), log.value.decode(errors="replace")))
        uk = type("UserKernel", (), {})()
        uk.ptr, uk.d, uk.n_ep, uk.has_valid, uk.log = ptr, int(d), int(n_ep), "QSMC_USER_HAS_VALID" in source, log.value.decode(errors="replace")
        cache[key] = uk
        return uk

    def update_user(self, uk, x, w_in, w_out, prev_norm, ep_vec, outcome, moments=False):
        """`update_fused` for a compiled user model (qsmc_update_user); ep_vec: the experiment as n_ep float64."""
        if self._design_jobs:
            self._no_design_in_flight("update_user")
        d = x.shape[0]
        self.update_gen += 1
        ep_vec = np.ascontiguousarray(ep_vec, dtype=np.float64)
        self._chk(self.lib.qsmc_update_user(
            self.h, uk.ptr, x.data_ptr(), x.stride(0), x.shape[1], w_in.data_ptr() if w_in is not None else None,
            w_out.data_ptr(), float(prev_norm), ep_vec.ctypes.data_as(C.POINTER(C.c_double)), int(outcome),
            self._stats.data_ptr(), self._st_ref, self._mom_ptr[d] if moments else None, self.stream()), "qsmc_update_user")
        return self._st

    def update_multi_user(self, uk, x, w_in, w_out, prev_norm, eps, outcomes):
        """`update_multi` for a compiled user model (qsmc_update_multi_user): eps (k, n_ep) float64, k <= 8 data."""
        if self._design_jobs:
            self._no_design_in_flight("update_multi_user")
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        k, d = eps.shape[0], x.shape[0]
        oc = (C.c_int64 * k)(*[int(o) for o in outcomes])
        st = (_native.UpdateStats * k)()
        mom = np.empty(d + d * (d + 1) // 2, dtype=np.float64) if d <= 4 else None
        flat = np.ascontiguousarray(eps.reshape(-1)) if eps.size else np.zeros(1)
        self._chk(self.lib.qsmc_update_multi_user(
            self.h, uk.ptr, self._p(x), x.stride(0), x.shape[1], self._p(w_in) if w_in is not None else None, self._p(w_out),
            float(prev_norm), flat.ctypes.data_as(C.POINTER(C.c_double)), oc, k, st,
            _native.f64_ptr(mom) if mom is not None else None, self.stream()), "qsmc_update_multi_user")
        self.update_gen += 1
        if mom is None:
            return list(st), None, None
        return list(st), mom[:d].copy(), self._unpack_upper(mom[d:], d)

    def likelihood_user(self, uk, x, eps, outcomes):
        """L[n_o, n_e, N] on the device from a compiled user model; eps: (n_e, n_ep) float64."""
        n = x.shape[1]
        eps = np.ascontiguousarray(eps, dtype=np.float64)
        if eps.ndim != 2 or eps.shape[1] != uk.n_ep:
            raise ValueError("likelihood_user: experiments must be an (n_e, {}) array".format(uk.n_ep))
        n_e, n_o = eps.shape[0], len(outcomes)
        out = self.empty(n_o, n_e, n)
        if n == 0 or n_e == 0 or n_o == 0:
            return out
        oc = (C.c_int64 * n_o)(*[int(o) for o in outcomes])
        flat = np.ascontiguousarray(eps.reshape(-1)) if eps.size else np.zeros(1)
        self._chk(self.lib.qsmc_likelihood_user(self.h, uk.ptr, self._p(x), x.stride(0), n,
                                                flat.ctypes.data_as(C.POINTER(C.c_double)), n_e, oc, n_o, self._p(out),
                                                self.stream()), "qsmc_likelihood_user")
        return out

    def valid_user(self, uk, x):
        """uint8 device mask [n] from a compiled user model's valid() (all ones if it defines none); x: (d, n) SoA, any
        row stride, unit column stride."""
        if x.stride(1) != 1:
            x = x.contiguous()
        n = x.shape[1]
        out = self.empty(n, dtype=self.torch.uint8)
        if n:
            self._chk(self.lib.qsmc_valid_user(self.h, uk.ptr, self._p(x), x.stride(0), n, self._p(out), self.stream()),
                      "qsmc_valid_user")
        return out

    def update_from_likelihood(self, L, w_in, w_out, prev_norm):
        st = _native.UpdateStats()
        self._chk(self.lib.qsmc_update_from_likelihood(
            self.h, self._p(L), w_in.shape[0], self._p(w_in), self._p(w_out), float(prev_norm),
            self._p(self._stats), C.byref(st), self.stream()), "qsmc_update_from_likelihood")
        return st

    def clip_weights(self, w, norm):
        st = _native.UpdateStats()
        self._chk(self.lib.qsmc_clip_weights(self.h, self._p(w), w.shape[0], float(norm),
                                             self._p(self._stats), C.byref(st), self.stream()),
                  "qsmc_clip_weights")
        return st

    def weight_stats(self, w, norm):
        st = _native.UpdateStats()
        self._chk(self.lib.qsmc_weight_stats(self.h, self._p(w), w.shape[0], float(norm),
                                             self._p(self._stats), C.byref(st), self.stream()),
                  "qsmc_weight_stats")
        return st

    def normalized_weights(self, w, norm):
        out = self.empty(w.shape[0])
        self._chk(self.lib.qsmc_normalize_weights(self.h, self._p(w), self._p(out), w.shape[0],
                                                  float(norm), self.stream()), "qsmc_normalize_weights")
        return out

    def argsort(self, keys, descending=False):
        """(sorted keys, permutation as int64) of a 1-D float64 device tensor (device radix sort)."""
        n = keys.shape[0]
        out = self.empty(n)
        idx = self.empty(n, dtype=self.torch.int64)
        self._chk(self.lib.qsmc_argsort(self.h, self._p(keys), n, int(bool(descending)), self._p(out), self._p(idx),
                                        self.stream()), "qsmc_argsort")
        return out, idx

    def searchsorted(self, a, q, side="left"):
        """Device table a (non-decreasing), host or device queries q -> int64 device tensor of insertion points."""
        if not isinstance(q, self.torch.Tensor):
            q = self.to_device(np.asarray(q, dtype=np.float64))
        out = self.empty(q.shape[0], dtype=self.torch.int64)
        self._chk(self.lib.qsmc_searchsorted(self.h, self._p(a), a.shape[0], self._p(q), q.shape[0],
                                             0 if side == "left" else 1, self._p(out), self.stream()),
                  "qsmc_searchsorted")
        return out

    def gather_rows(self, x, idx):
        """x[:, idx] for a (d, n) device tensor and int64 device indices (the Liu-West centre kernel with a = 1)."""
        return self.lw_centres(x, idx, 1.0, np.zeros(x.shape[0]))

    def weight_entropy(self, w, n, norm):
        out = C.c_double()
        self._chk(self.lib.qsmc_weight_entropy(self.h, self._p(w) if w is not None else None, int(n), float(norm),
                                               C.byref(out), self.stream()), "qsmc_weight_entropy")
        return out.value

    def kde_cross_entropy(self, x, w, norm_p, y, v, norm_q, scale):
        """sum_i p_i log sum_j q_j phi(||scale o (x_i - y_j)||): x (d, n), y (d, m) device SoA clouds, w / v their
        unnormalised weights (None = all ones) with normalisers norm_p / norm_q, scale = sqrt(Q) / delta (host, d)."""
        d = x.shape[0]
        scale = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), (d,)))
        out = C.c_double()
        self._chk(self.lib.qsmc_kde_cross_entropy(
            self.h, self._p(x), x.stride(0), x.shape[1], self._p(w) if w is not None else None, float(norm_p),
            self._p(y), y.stride(0), y.shape[1], self._p(v) if v is not None else None, float(norm_q), d,
            _native.f64_ptr(scale), C.byref(out), self.stream()), "qsmc_kde_cross_entropy")
        return out.value

    def normalize_weights_into(self, w_in, w_out, norm):
        """w_out = w_in / norm (may alias)."""
        self._chk(self.lib.qsmc_normalize_weights(self.h, self._p(w_in), self._p(w_out), w_in.shape[0],
                                                  float(norm), self.stream()), "qsmc_normalize_weights")

    def fill(self, w, value):
        self._chk(self.lib.qsmc_fill(self.h, self._p(w), w.shape[0], float(value), self.stream()),
                  "qsmc_fill")

    # ------------------------------------------------------------------ RCCL transport (sharded updater)
    @staticmethod
    def comm_unique_id():
        """128-byte RCCL id (rank 0 creates it; broadcast to the other ranks by the caller)."""
        buf = C.create_string_buffer(128)
        _native.check(None, _native.load().qsmc_comm_unique_id(buf), "qsmc_comm_unique_id")
        return buf.raw

    def comm_init(self, rank, nranks, unique_id):
        """Collective: every rank of the shard group, same id."""
        self._chk(self.lib.qsmc_comm_init(self.h, int(rank), int(nranks), C.create_string_buffer(unique_id, 128)),
                  "qsmc_comm_init")
        self._cc_tot = np.empty(self.REDUCE_MAX)          # the C side's bound on n (+ nranks) -- see allreduce_sums
        self._cc_first = np.empty(int(nranks))
        self._cc_nranks = int(nranks)

    REDUCE_MAX = 188              # REDUCE_OUT_MAX - 4 reserved tail slots of the pinned block (qsmc_kernels.hip)

    def comm_count(self):
        """(ranks in the communicator, this rank's index) as RCCL reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int32(), C.c_int32()
        self._chk(self.lib.qsmc_comm_count(self.h, C.byref(n), C.byref(r)), "qsmc_comm_count")
        return n.value, r.value

    def comm_destroy(self):
        self._chk(self.lib.qsmc_comm_destroy(self.h), "qsmc_comm_destroy")

    def allreduce_sums(self, vec_dev, n, min_index=-1):
        """RCCL all-gather + rank-ordered sum of the first n doubles of a device vector on the launch stream.  Returns
        (tot, firsts): views of reused host arrays -- sums over ranks (entry min_index: minimum) and every rank's
        entry 0."""
        if n + self._cc_nranks > self.REDUCE_MAX:
            raise ValueError("allreduce_sums: n + nranks = {} exceeds {}".format(n + self._cc_nranks, self.REDUCE_MAX))
        self._chk(self.lib.qsmc_allreduce_sums(self.h, self._p(vec_dev), int(n), int(min_index), self._cc_tot.ctypes.data,
                                               self._cc_first.ctypes.data, self.stream()), "qsmc_allreduce_sums")
        return self._cc_tot[:n], self._cc_first

    def publish_rows(self, rows_dev, n, nranks, min_index=-1):
        """The device half of `allreduce_sums` on caller-supplied rows ([nranks][n] doubles on the device): (tot, firsts)
        -- the rank-ordered sums (entry min_index: the minimum) and every rank's entry 0.  No communicator involved."""
        tot = np.empty(n, dtype=np.float64)
        firsts = np.empty(nranks, dtype=np.float64)
        self._chk(self.lib.qsmc_publish_rows(self.h, self._p(rows_dev), int(n), int(min_index), int(nranks),
                                             _native.f64_ptr(tot), _native.f64_ptr(firsts), self.stream()), "qsmc_publish_rows")
        return tot, firsts

    # ------------------------------------------------------------------ moments
    def moments(self, x, w, norm):
        """Returns host (sum_w, S1[d], S2[d, d]) of the normalised weights."""
        if self._design_jobs:
            self._no_design_in_flight("moments")
        d, n = x.shape
        if d > _native.QSMC_MAX_D_WIDE:
            # beyond the library's kernels (a plugin model with more than 64 parameters, e.g. four-qubit tomography): the
            # same sums as torch products on the device -- (d x N)(N x d) on the matrix cores through rocBLAS
            t = self.torch
            wn = (t.ones(n, dtype=t.float64, device=x.device) if w is None else w) / float(norm)
            s0 = float(wn.sum().item())
            xw = x * wn
            return s0, xw.sum(dim=1).cpu().numpy(), (xw @ x.T).cpu().numpy()
        k = 1 + d + d * (d + 1) // 2
        out = np.empty(k, dtype=np.float64)
        self._chk(self.lib.qsmc_moments(self.h, self._p(x), x.stride(0), n, d, self._p(w), float(norm),
                                        None, _native.f64_ptr(out), self.stream()), "qsmc_moments")
        return float(out[0]), out[1:1 + d].copy(), self._unpack_upper(out[1 + d:], d)

    def sqrtm_psd(self, A, scale=1.0):
        """(scale * sqrtm_psd(A), || sqrt sqrt - A ||_F): utils.py:593-607.  d <= 16: the library's round-robin Jacobi (the
        routine qsmc_step and the device wavefront run: one set of bits); wide clouds (d > 16): LAPACK's eigh through NumPy --
        the C loop takes 2.2 ms at d = 64 with the GPU waiting for S, eigh 0.2 -- to the same tolerance (nothing compares the
        wide square root bit for bit)."""
        A = np.ascontiguousarray(A, dtype=np.float64)
        d = A.shape[0]
        if d > _native.QSMC_MAX_D:
            try:
                lam, V = np.linalg.eigh(0.5 * (A + A.T))
            except np.linalg.LinAlgError:
                return np.full((d, d), np.nan), float("inf")
            sq = (V * np.sqrt(np.maximum(lam, 0.0))) @ V.T
            return float(scale) * sq, float(np.linalg.norm(sq @ sq - A))
        S = np.empty((d, d))
        err = C.c_double()
        self._chk(self.lib.qsmc_sqrtm_psd(_native.f64_ptr(A), d, float(scale), _native.f64_ptr(S),
                                          C.byref(err)), "qsmc_sqrtm_psd")
        return S, err.value

    # ------------------------------------------------------------------ Liu-West pieces
    def cumsum(self, w, norm):
        cdf = self.empty(w.shape[0])
        self._chk(self.lib.qsmc_cumsum(self.h, self._p(w), w.shape[0], float(norm), self._p(cdf),
                                       self.stream()), "qsmc_cumsum")
        return cdf

    def lw_ancestors(self, cdf, u):
        js = self.empty(u.shape[0], dtype=self.torch.int64)
        self._chk(self.lib.qsmc_lw_ancestors(self.h, self._p(cdf), cdf.shape[0], self._p(u), u.shape[0],
                                             self._p(js), self.stream()), "qsmc_lw_ancestors")
        return js

    def lw_centres(self, x_in, js, a, mean):
        d = x_in.shape[0]
        if d > _native.QSMC_MAX_D_WIDE:        # (beyond the kernels: a plugin model with more than 64 parameters -- torch, on the device)
            return float(a) * x_in[:, js] + (1.0 - float(a)) * self.to_device(np.asarray(mean, dtype=np.float64))[:, None]
        mus = self.empty(d, js.shape[0])
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        self._chk(self.lib.qsmc_lw_centres(self.h, self._p(x_in), x_in.stride(0), d, self._p(js),
                                           js.shape[0], float(a), _native.f64_ptr(mean), self._p(mus),
                                           mus.stride(0), self.stream()), "qsmc_lw_centres")
        return mus

    def lw_perturb(self, desc, postselect, mus, idxs, k, centre_by_idx, S, z, x_out):
        if mus.shape[0] > _native.QSMC_MAX_D_WIDE:
            # the same round in torch (resamplers.py:325-338): x_i = mu_i + S z_i for the k particles still to be drawn; the
            # model's own validity test decides (the caller's: no kernel knows a model of this size)
            kick = self.to_device(np.asarray(S, dtype=np.float64)) @ z[:, :k]
            centre = mus[:, :k] if (idxs is None or not centre_by_idx) else mus[:, idxs]
            if idxs is None:
                x_out[:, :k] = centre + kick
            else:
                x_out[:, idxs] = centre + kick
            return self.torch.ones(k, dtype=self.torch.uint8, device=x_out.device)
        valid = self.empty(k, dtype=self.torch.uint8)
        S = np.ascontiguousarray(S, dtype=np.float64)
        self._chk(self.lib.qsmc_lw_perturb(
            self.h, C.byref(desc), int(bool(postselect)), self._p(mus), mus.stride(0),
            self._p(idxs) if idxs is not None else None, k, int(bool(centre_by_idx)), _native.f64_ptr(S),
            self._p(z), z.stride(0), self._p(x_out), x_out.stride(0), self._p(valid), self.stream()),
            "qsmc_lw_perturb")
        return valid

    def lw_resample_philox(self, desc, postselect, x_in, w, norm, a, mean, S, n_out, seed, epoch, maxiter,
                           sync=True, out=None, canon=None, expect_redraws=0):
        """Returns (x_out, n_failed); with sync=False n_failed is None and the count is available
        from `last_resample_failed()` after the next stream synchronisation.  `out`: a (d, n_out) device
        view to fill (row stride arbitrary) instead of a fresh tensor; x_in may be a column slice of a cloud.
        `canon` = (kind, basis_dev, allow_subnormalized) from TomographyModel._native_canonicalize_fused: the new cloud
        comes out canonicalized (qsmc_lw_fuse_canonicalize)."""
        if self._design_jobs:
            self._no_design_in_flight("lw_resample_philox")
        d = x_in.shape[0]
        if expect_redraws:
            # (how many first tries of this cloud's previous resample failed postselection: the library then banks
            #  spare proposals for that many redraws -- qsmc_lw_expect_redraws)
            self._chk(self.lib.qsmc_lw_expect_redraws(self.h, int(expect_redraws)), "qsmc_lw_expect_redraws")
        if canon is not None:
            kind, basis_dev, allow_sub = canon
            self._chk(self.lib.qsmc_lw_fuse_canonicalize(self.h, self._p(basis_dev) if basis_dev is not None else None, 4,
                                                         1 if kind == 1 else 0, int(bool(allow_sub))),
                      "qsmc_lw_fuse_canonicalize")
        x_out = self.empty(d, n_out) if out is None else out
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        S = np.ascontiguousarray(S, dtype=np.float64)
        failed = C.c_int64()
        self._chk(self.lib.qsmc_lw_resample_philox(
            self.h, C.byref(desc), int(bool(postselect)), self._p(x_in), x_in.stride(0), x_in.shape[1], d,
            self._p(w) if w is not None else None, float(norm), float(a), _native.f64_ptr(mean),
            _native.f64_ptr(S), n_out, C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(epoch), int(maxiter),
            self._p(x_out), x_out.stride(0), C.byref(failed) if sync else None, self.stream()),
            "qsmc_lw_resample_philox")
        return x_out, (failed.value if sync else None)

    def fused_canon_applies(self, d, n_in, n_out):
        """Does a resample of this shape take the split d = 16 sampler (the one that can fold canonicalize in)?  The
        library's own rule, asked of the library (qsmc_lw_can_fuse_canonicalize)."""
        return bool(self.lib.qsmc_lw_can_fuse_canonicalize(int(d), int(n_in), int(n_out)))

    def reserve(self, n_in, n_out, d):
        """Grow the handle's update / resample scratch for a cloud of this shape now, not inside the first resample."""
        self._chk(self.lib.qsmc_reserve(self.h, int(n_in), int(n_out), int(d)), "qsmc_reserve")

    def step_sqrt_stats(self):
        """(d = 16 square roots formed on the device by qsmc_step, of those confirmed bit for bit by the host)."""
        q, a = C.c_int64(), C.c_int64()
        self._chk(self.lib.qsmc_step_sqrt_stats(self.h, C.byref(q), C.byref(a)), "qsmc_step_sqrt_stats")
        return q.value, a.value

    def random_walk(self, x, scale, z=None, seed=0, epoch=0):
        """x[m, :] += scale[m] * z in place (rows with scale 0 untouched); z: device (n_rw, n) steps, or None
        for Philox standard normals."""
        scale = np.ascontiguousarray(scale, dtype=np.float64)
        self._chk(self.lib.qsmc_random_walk(
            self.h, self._p(x), x.stride(0), x.shape[1], x.shape[0], _native.f64_ptr(scale),
            self._p(z) if z is not None else None,
            (z.stride(0) if z.shape[0] > 1 else z.shape[1]) if z is not None else 0,
            C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(epoch), self.stream()), "qsmc_random_walk")

    def use_update_sums(self, token):
        """Vouch that the weights about to be resampled are the untouched output of update number `token`."""
        self._chk(self.lib.qsmc_lw_use_update_sums(self.h, C.c_uint64(int(token))), "qsmc_lw_use_update_sums")

    def lw_resample_prepare(self, w, n_in, norm, n_out, seed, epoch):
        """Queue the weight-only prefix of the next `lw_resample_philox` call with the same arguments."""
        self._chk(self.lib.qsmc_lw_resample_prepare(
            self.h, self._p(w) if w is not None else None, int(n_in), float(norm), int(n_out),
            C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(epoch), self.stream()), "qsmc_lw_resample_prepare")

    def arm_resample_prefix(self, key):
        """key = (ess_below, n_out, seed, epoch) or None: from now on every host-visible `update_fused` queues the
        resampler's weight-only prefix for that resample behind itself, gated on the device-side ESS test
        (qsmc_lw_arm_prefix).  Called only when the key changes -- once per resample."""
        if key == self._armed_prefix:
            return
        self._armed_prefix = key
        if key is None:
            self._chk(self.lib.qsmc_lw_arm_prefix(self.h, 0, 0.0, 1, C.c_uint64(0), C.c_uint64(0)), "qsmc_lw_arm_prefix")
        else:
            self._chk(self.lib.qsmc_lw_arm_prefix(self.h, 1, float(key[0]), int(key[1]),
                                                  C.c_uint64(key[2] & (2 ** 64 - 1)), C.c_uint64(key[3])),
                      "qsmc_lw_arm_prefix")

    def last_resample_redraws(self):
        """Outputs of the latest resample that needed a global redraw (known after the next synchronising call)."""
        out = C.c_int64()
        self._chk(self.lib.qsmc_last_resample_redraws(self.h, C.byref(out)), "qsmc_last_resample_redraws")
        return out.value

    def prefix_stats(self):
        """(speculative prefixes queued, resamples that found theirs done) on this handle."""
        q, a = C.c_int64(), C.c_int64()
        self._chk(self.lib.qsmc_lw_prefix_stats(self.h, C.byref(q), C.byref(a)), "qsmc_lw_prefix_stats")
        return q.value, a.value

    def last_resample_failed(self, synchronize=False):
        out = C.c_int64()
        self._chk(self.lib.qsmc_last_resample_failed(self.h, C.byref(out), int(bool(synchronize)), self.stream()),
                  "qsmc_last_resample_failed")
        return out.value

    def lw_resample_philox_sharded(self, desc, postselect, x_in, w, norm, a, mean, S, dest_counts, seed, epoch,
                                   maxiter, sync=True):
        """Finished particles for every destination rank: AoS rows (sum(dest_counts), d), grouped by rank."""
        d = x_in.shape[0]
        counts = np.ascontiguousarray(dest_counts, dtype=np.int64)
        n_out = int(counts.sum())
        rows = self.empty(n_out, d)
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        S = np.ascontiguousarray(S, dtype=np.float64)
        failed = C.c_int64()
        self._chk(self.lib.qsmc_lw_resample_philox_sharded(
            self.h, C.byref(desc), int(bool(postselect)), self._p(x_in), x_in.stride(0), x_in.shape[1], d,
            self._p(w) if w is not None else None, float(norm), float(a), _native.f64_ptr(mean),
            _native.f64_ptr(S), counts.ctypes.data_as(C.POINTER(C.c_int64)), len(counts),
            C.c_uint64(seed & (2 ** 64 - 1)),
            C.c_uint64(epoch), int(maxiter), self._p(rows), C.byref(failed) if sync else None, self.stream()),
            "qsmc_lw_resample_philox_sharded")
        return rows, (failed.value if sync else None)

    def prior_uniform_philox(self, desc, postselect, lo, hi, n, seed, epoch, maxiter=100):
        d = len(lo)
        x_out = self.empty(d, n)
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        failed = C.c_int64()
        self._chk(self.lib.qsmc_prior_uniform_philox(
            self.h, C.byref(desc), int(bool(postselect)), _native.f64_ptr(lo), _native.f64_ptr(hi), d, n,
            C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(epoch), int(maxiter), self._p(x_out),
            x_out.stride(0), C.byref(failed), self.stream()), "qsmc_prior_uniform_philox")
        return x_out, failed.value

    def tomo_canonicalize(self, basis_dev, dim, x, allow_subnormalized, pauli=False):
        """`pauli`: the basis is the reference's n-qubit Pauli basis (sparse contraction for 2 qubits)."""
        self._chk(self.lib.qsmc_tomo_canonicalize2(self.h, self._p(basis_dev), dim, 1 if pauli else 0, self._p(x),
                                                   x.stride(0), x.shape[1], int(bool(allow_subnormalized)), self.stream()),
                  "qsmc_tomo_canonicalize2")
