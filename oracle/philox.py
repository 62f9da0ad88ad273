"""NumPy emulation of the device Philox4x32-10 streams (TEST INFRASTRUCTURE).

Mirrors `PhiloxStream` in python-qinfer_amd/csrc/qsmc_device.h bit for bit so the device-RNG
resample / prior kernels can be checked against the oracle on identical random numbers.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)) & MASK, lo1, (hi0 ^ c3 ^ np.uint64(k1)) & MASK, lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u53(a, b):
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 +
            (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


def keys(seed, epoch):
    seed = int(seed) & (2 ** 64 - 1)
    return seed & 0xFFFFFFFF, ((seed >> 32) ^ (int(epoch) >> 16)) & 0xFFFFFFFF


def uniforms(particles, seed, epoch, rnd, slot, key_xor=(0, 0)):
    """(u0, u1) for each particle index at (epoch, round, slot).  key_xor: domain separation applied to
    the two key words (qsmc_random_walk keeps its streams apart from the resampler's this way)."""
    particles = np.asarray(particles, dtype=np.uint64)
    k0, k1 = keys(seed, epoch)
    k0, k1 = k0 ^ key_xor[0], k1 ^ key_xor[1]
    er = np.uint64(((int(epoch) & 0xFFFF) << 16) | int(rnd))
    r = philox4x32_10(particles & MASK, particles >> np.uint64(32), np.full(particles.shape, er),
                      np.full(particles.shape, np.uint64(slot)), k0, k1)
    return u53(r[0], r[1]), u53(r[2], r[3])


WALK_KEY_XOR = (0x52574B31, 0x9E3779B9)


def random_walk_normals(n, n_rw, seed, epoch):
    """Standard normals of qsmc_random_walk (z == NULL): (n_rw, n); particles 2P and 2P + 1 share block
    (P, epoch, slot r) of walking parameter r -- Box-Muller component i & 1."""
    i = np.arange(n, dtype=np.int64)
    out = np.empty((n_rw, n))
    for r in range(n_rw):
        za, zb = normals(i >> 1, seed, epoch, 0, r, key_xor=WALK_KEY_XOR)
        out[r] = np.where(i & 1, zb, za)
    return out


def normals(particles, seed, epoch, rnd, slot, key_xor=(0, 0)):
    u0, u1 = uniforms(particles, seed, epoch, rnd, slot, key_xor)
    r = np.sqrt(-2.0 * np.log(1.0 - u0))
    return r * np.cos(2 * np.pi * u1), r * np.sin(2 * np.pi * u1)


def liu_west_philox(w, x, valid_fn, a, h, seed, epoch, n_out, maxiter=1000, postselect=True,
                    mean=None, cov=None, zero_cov_comp=1e-10):
    """Oracle of qsmc_lw_resample_philox: same draws, same redraw rule (ancestor AND kick)."""
    import np_oracle as orc
    N, d = x.shape
    mean = orc.particle_mean(w, x) if mean is None else mean
    cov = orc.particle_cov(w, x, warn=False) if cov is None else cov
    if np.linalg.norm(cov, 'fro') == 0:
        cov = zero_cov_comp * np.eye(d)
    S = h * orc.sqrtm_psd(cov)[0]
    cdf = np.cumsum(w)
    out = np.empty((n_out, d))
    todo = np.arange(n_out)
    for rnd in range(maxiter):
        if not todo.size:
            break
        u, _ = uniforms(todo, seed, epoch, rnd, 0)
        js = np.minimum(cdf.searchsorted(u, side='right'), N - 1)
        z = np.empty((d, todo.size))
        for q in range(0, d, 2):
            z0, z1 = normals(todo, seed, epoch, rnd, 1 + q // 2)
            z[q] = z0
            if q + 1 < d:
                z[q + 1] = z1
        out[todo] = (a * x[js] + (1 - a) * mean) + (S @ z).T
        ok = valid_fn(out[todo]) if postselect else np.ones(todo.size, dtype=bool)
        todo = todo[~ok]
    return out, todo.size


BUCKET_CHUNK = 4096


def _pair_word(ids, seed, epoch, slot, rnd=0):
    """word (id & 1) of Philox block (id >> 1, round `rnd`, slot): two outputs share one block."""
    ids = np.asarray(ids, dtype=np.int64)
    ua, ub = uniforms(ids >> 1, seed, epoch, rnd, slot)
    return np.where(ids & 1, ub, ua)


def _pair_normal(n_idx, seed, epoch, slot, rnd=0):
    """Box-Muller component (n & 1) of block (n >> 1, round `rnd`, slot)."""
    n_idx = np.asarray(n_idx, dtype=np.int64)
    za, zb = normals(n_idx >> 1, seed, epoch, rnd, slot)
    return np.where(n_idx & 1, zb, za)


_STIRLING_SMALL = (0.08106146679532726, 0.0413406959554093, 0.02767792568499834, 0.020790672103765093,
                   0.016644691189821193, 0.013876128823070748, 0.01189670994589177, 0.010411265261972096,
                   0.009255462182712733, 0.00833056343336287)


def _stirling_tail(k):
    """ln k! - [(k + 1/2) ln(k + 1) - (k + 1) + ln(2 pi) / 2]: table for k < 10, five terms of the series beyond."""
    if k < 10.0:
        return _STIRLING_SMALL[int(k)]
    rx = 1.0 / (k + 1.0)
    r2 = rx * rx
    return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0 - (1.0 / 1680.0 - 1.0 / 1188.0 * r2) * r2) * r2) * r2) * rx


def poisson_draw(mu, node, seed, epoch, rnd=0):
    """X ~ Poisson(mu) exactly as poisson_draw of the device library (csrc/kernels/resample.hpp): sequential search of
    the cdf for mu < 10, PTRS (W. Hoermann, Insurance: Mathematics and Economics 12 (1993) 39: transformed
    rejection with squeeze) otherwise; attempt t of chunk `node` consumes Philox block (node | t << 32, round 0,
    slot 0) and the first accepted attempt is the draw.  Scalar floats in the device's order of operations."""
    mu = float(mu)
    if not (mu > 0.0):
        return 0
    f = float

    def block(t):
        u, v = uniforms(np.array([node | (t << 32)], dtype=np.int64), seed, epoch, rnd, 0)
        return f(u[0]), f(v[0])

    def ln(x):
        return f(np.log(x)) if x > 0.0 else -np.inf

    if mu < 10.0:
        U, _ = block(0)
        pk = f(np.exp(-mu))
        cdf, X = pk, 0.0
        while U > cdf and X < 200.0:
            X += 1.0
            pk = pk * mu / X
            cdf += pk
        return int(X)
    smu = f(np.sqrt(mu))
    lmu = ln(mu)
    b = 0.931 + 2.53 * smu
    a = -0.059 + 0.02483 * b
    linva = ln(1.1239 + 1.1328 / (b - 3.4))
    vr = 0.9277 - 3.6224 / (b - 2.0)
    for t in range(4096):
        U, V = block(t)
        u = U - 0.5
        us = 0.5 - abs(u)
        if us == 0.0:
            continue                                             # (k = -inf on the device: rejected)
        kk = f(np.floor((2.0 * a / us + b) * u + mu + 0.43))
        if us >= 0.07 and V <= vr:
            return int(kk)
        if kk < 0.0 or (us < 0.013 and V > us):
            continue
        lhs = ln(V) + linva - ln(a / (us * us) + b)
        lgk = (kk + 0.5) * ln(kk + 1.0) - (kk + 1.0) + 0.91893853320467274178 + _stirling_tail(kk)
        if lhs <= -mu + kk * lmu - lgk:
            return int(kk)
    return 0


def poissonised_counts(edges, n_out, seed, epoch, margin=5.0):
    """Multinomial(n_out; chunk masses) the device's way (k_bucket_counts): independent
    Poisson((n_out - margin sqrt(n_out)) p_c) per chunk, then the shortfall as categorical draws against the chunk
    edges (word j & 1 of Philox block (j >> 1, round 0, slot 3)) -- or, should the Poisson total overshoot, the
    surplus removed item by item uniformly at random (word 0 of block (i, round 0, slot 4)).
    edges[c] = upper CDF edge of chunk c."""
    edges = np.asarray(edges, dtype=np.float64)
    chunks = len(edges)
    lam = max(float(n_out) - margin * float(np.sqrt(float(n_out))), 0.0)
    lo = np.concatenate([[0.0], edges[:-1]])
    mass = edges - lo
    total = float(edges[-1])
    counts = np.zeros(chunks, dtype=np.int64)
    for c in range(chunks):
        if mass[c] > 0.0 and total > 0.0:
            counts[c] = poisson_draw(lam * float(mass[c]) / total, c, seed, epoch)
    T = int(counts.sum())
    if T < n_out:
        u = _pair_word(np.arange(n_out - T), seed, epoch, 3)
        c_of = np.minimum(np.searchsorted(edges, u, side='right'), chunks - 1)
        counts += np.bincount(c_of, minlength=chunks)
    else:
        left = T
        for i in range(T - n_out):
            u, _ = uniforms(np.array([i], dtype=np.int64), seed, epoch, 0, 4)
            target = min(int(float(u[0]) * float(left)), left - 1)
            c = int(np.searchsorted(np.cumsum(counts), target, side='right'))
            counts[c] -= 1
            left -= 1
    return counts


def bucket_cap(n_out):
    """Outputs per work item (qsmc_kernels.hip: bucket_cap): largest power of two <= 8192 leaving >= 1024 items."""
    cap = 2 * BUCKET_CHUNK
    while cap > 512 and n_out // cap < 1024:
        cap >>= 1
    return cap


BANK_ROUNDS, BANK_MAX_PER_ITEM = 7, 4096
_M64 = (1 << 64) - 1


def bank_keys(seed, epoch):
    """The four keys of the bank's bijection (qsmc_kernels.hip: bank_layout): splitmix64 from seed and epoch."""
    x = (int(seed) ^ ((int(epoch) * 0xD1342543DE82EF95) & _M64) ^ 0x62616E6B) & _M64
    keys = []
    for _ in range(4):
        x = (x + 0x9E3779B97F4A7C15) & _M64
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
        keys.append(z ^ (z >> 31))
    return keys


def bank_perm(g, n, key):
    """kernels/resample.hpp: bank_perm -- a keyed bijection of [0, n), cycle-walked on ceil(log2 n) bits."""
    if n < 2:
        return 0
    b = 1
    while (1 << b) < n:
        b += 1
    mask = (1 << b) - 1
    s1, s2 = max(b // 2, 1), max((b + 2) // 3, 1)
    x = int(g)
    while True:
        x = (x * 0x9E3779B97F4A7C15 + key[0]) & mask
        x ^= x >> s1
        x = (x * 0xBF58476D1CE4E5B9 + key[1]) & mask
        x ^= x >> s2
        x = (x * 0x94D049BB133111EB + key[2]) & mask
        x ^= x >> s1
        x = (x * 0xD6E8FEB86659FD93 + key[3]) & mask
        x ^= x >> s2
        if x < n:
            return x


def bank_spares(x, cdf, edges, counts, cap, lam, a, mean, S, valid_fn, seed, epoch):
    """The proposal bank as k_bucket_sample_ordered fills it: per work item (chunk, part) e_i ~ Poisson(lam mass / total /
    parts) spares (block (item, attempt), round tag 0xFFFE); spare k of item i: position from word k & 1 of block
    ((i << 12) | k >> 1, round tag 0xFFFF, slot 1), normals from the same pair id (slot 2); spare k is kicked from its own
    ancestor (drawing order: unlike the primaries the spares are not sorted by ancestor -- round 5).
    Returns (values (E, d), valid (E,)) in the bank's logical order: items ascending, spares in order."""
    N, d = x.shape
    chunks = len(edges)
    total = float(edges[-1])
    vals, oks = [], []
    item = 0
    for c in range(chunks):
        parts = -(-int(counts[c]) // cap)
        lo = 0.0 if c == 0 else float(edges[c - 1])
        hi = float(edges[c])
        for _ in range(parts):
            mu = lam * ((hi - lo) / total) / float(parts) if (total > 0.0 and hi > lo) else 0.0
            e_i = min(poisson_draw(mu, item, seed, epoch, rnd=0xFFFE), BANK_MAX_PER_ITEM)
            if e_i:
                ids = 2 * ((item << 12) + (np.arange(e_i) >> 1)) + (np.arange(e_i) & 1)
                u = lo + _pair_word(ids, seed, epoch, 1, rnd=0xFFFF) * (hi - lo)
                base, end = c * BUCKET_CHUNK, min((c + 1) * BUCKET_CHUNK, N)
                js = np.minimum(np.maximum(np.searchsorted(cdf, u, side='right'), base), end - 1)
                z = np.stack([_pair_normal(ids * d + q, seed, epoch, 2, rnd=0xFFFF) for q in range(d)])
                v = (a * x[js] + (1 - a) * mean) + (S @ z).T
                vals.append(v)
                oks.append(valid_fn(v))
            item += 1
    if not vals:
        return np.zeros((0, d)), np.zeros((0,), dtype=bool)
    return np.concatenate(vals), np.concatenate(oks)


def bank_serve(failed_slots, bank_vals, bank_ok, out, seed, epoch):
    """k_bank_round: round t hands the j-th slot still failed (ascending slot order, order kept from round to round)
    spare perm(B_t + j).  Returns the slots left for the global-CDF redraw."""
    E = bank_vals.shape[0]
    key = bank_keys(seed, epoch)
    todo = np.sort(np.asarray(failed_slots, dtype=np.int64))
    leftover = []
    B = 0
    for _ in range(BANK_ROUNDS):
        nxt = []
        for j, slot in enumerate(todo):
            g = B + j
            if g >= E:
                leftover.append(int(slot))
                continue
            pg = bank_perm(g, E, key)
            if bank_ok[pg]:
                out[slot] = bank_vals[pg]
            else:
                nxt.append(int(slot))
        B += len(todo)
        todo = np.asarray(nxt, dtype=np.int64)
        if not todo.size:
            break
    leftover.extend(int(v) for v in todo)
    return np.asarray(leftover, dtype=np.int64)


def liu_west_philox_bucketed(w, x, valid_fn, a, h, seed, epoch, n_out, maxiter=1000, postselect=True,
                             mean=None, cov=None, zero_cov_comp=1e-10, cdf=None, margin=5.0, sort_items=None,
                             expect_redraws=0, z_stride=None):
    """Oracle of the bucketed device-RNG resampler (k_bucket_counts / k_bucket_sample).  Outputs are
    ordered by ancestor CHUNK.  Stream layout (round 0, two outputs per Philox block):
      slot 0: the Poisson chunk counts, slots 3 / 4 their top-up / removal (poissonised_counts); slot 1:
      within-chunk position of slot o (independent of the counts);
      slot 2: normal n = o * z_stride + q (z_stride = d; the wide kick kernel, 16 < d <= 64, pads it to a multiple of
      16: 16 ceil(d / 16), csrc/kernels/wide.hpp).  Retries (round r >= 1) are per output and redraw a GLOBAL
      ancestor from block (o, r, 0) and normals from (o, r, 1 + q // 2)."""
    import np_oracle as orc
    N, d = x.shape
    mean = orc.particle_mean(w, x) if mean is None else mean
    cov = orc.particle_cov(w, x, warn=False) if cov is None else cov
    if np.linalg.norm(cov, 'fro') == 0:
        cov = zero_cov_comp * np.eye(d)
    S = h * orc.sqrtm_psd(cov)[0]
    cdf = np.cumsum(w) if cdf is None else cdf
    chunks = (N + BUCKET_CHUNK - 1) // BUCKET_CHUNK
    edge_idx = np.minimum((np.arange(chunks) + 1) * BUCKET_CHUNK, N) - 1
    edges = cdf[edge_idx]
    ids = np.arange(n_out)
    counts = poissonised_counts(edges, n_out, seed, epoch, margin=margin)
    c_of_slot = np.repeat(np.arange(chunks), counts)                 # chunk of every output slot
    u_pos = _pair_word(ids, seed, epoch, 1)
    lo = np.where(c_of_slot == 0, 0.0, edges[np.maximum(c_of_slot - 1, 0)])
    hi = edges[c_of_slot]
    u = lo + u_pos * (hi - lo)
    base = c_of_slot * BUCKET_CHUNK
    end = np.minimum(base + BUCKET_CHUNK, N)
    js = np.minimum(np.maximum(np.searchsorted(cdf, u, side='right'), base), end - 1)
    if sort_items is None:
        sort_items = d >= 3
    if sort_items:
        # from d = 3 on the samplers (k_bucket_sample_ordered, k_bucket_sample16) kick the ancestors of a work item in
        # ascending order (neighbouring lanes then gather neighbouring particles): slot o_begin + k takes the k-th
        # smallest ancestor of the item and the normals of that slot -- the same law, the cloud being exchangeable
        cap = bucket_cap(n_out)
        slot0 = np.concatenate([[0], np.cumsum(counts)])
        for c in range(chunks):
            for b in range(int(slot0[c]), int(slot0[c + 1]), cap):
                e = min(b + cap, int(slot0[c + 1]))
                js[b:e] = np.sort(js[b:e], kind='stable')
    z_stride = d if z_stride is None else int(z_stride)
    z = np.stack([_pair_normal(ids * z_stride + q, seed, epoch, 2) for q in range(d)])      # (d, n_out)
    out = (a * x[js] + (1 - a) * mean) + (S @ z).T
    ok = valid_fn(out) if postselect else np.ones(n_out, dtype=bool)
    todo = ids[~ok]
    if expect_redraws > 0 and postselect and maxiter > 1 and d in (3, 4):
        # the proposal bank (qsmc_lw_expect_redraws): spares made by the sampler serve the failed first tries
        m = 1.25 * float(expect_redraws)
        lam = m + 6.0 * float(np.sqrt(m)) + 64.0
        bank_vals, bank_ok = bank_spares(x, cdf, edges, counts, bucket_cap(n_out), lam, a, mean, S, valid_fn, seed, epoch)
        todo = bank_serve(todo, bank_vals, bank_ok, out, seed, epoch)
    for rnd in range(1, maxiter):
        if not todo.size:
            break
        ur, _ = uniforms(todo, seed, epoch, rnd, 0)
        jr = np.minimum(cdf.searchsorted(ur, side='right'), N - 1)
        zr = np.empty((d, todo.size))
        for q in range(0, d, 2):
            z0, z1 = normals(todo, seed, epoch, rnd, 1 + q // 2)
            zr[q] = z0
            if q + 1 < d:
                zr[q + 1] = z1
        out[todo] = (a * x[jr] + (1 - a) * mean) + (S @ zr).T
        okr = valid_fn(out[todo]) if postselect else np.ones(todo.size, dtype=bool)
        todo = todo[~okr]
    return out, todo.size, js, counts
