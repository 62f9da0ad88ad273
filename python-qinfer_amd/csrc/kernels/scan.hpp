// kernels/scan.hpp -- inclusive scan of w / norm: chunk sums, scan of the chunk sums, chunk scans (qsmc_cumsum and the resampler's CDF).
// Part of the single translation unit qsmc_kernels.hip (included there, in this order; not a stand-alone header).
#pragma once

// =============================================================================================
// inclusive scan of w / norm  (three launches: chunk sums, scan of chunk sums, chunk scans)
// =============================================================================================
constexpr int SCAN_PER_LANE = 8;                                   // 8 consecutive particles per lane (64 B)
constexpr int SCAN_WAVE_CHUNK = QSMC_WAVE * SCAN_PER_LANE;         // 512 per wave
constexpr int SCAN_WAVES = 8;                                      // 512 threads scan one chunk
constexpr int SCAN_THREADS = SCAN_WAVES * QSMC_WAVE;
constexpr int SCAN_CHUNK = SCAN_WAVE_CHUNK * SCAN_WAVES;                // 4096 per workgroup

__device__ __forceinline__ double wave_inclusive_scan(double v, int lane) {
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const double t = __shfl_up(v, off, QSMC_WAVE);
        if (lane >= off) v += t;
    }
    return v;
}

// (the scan pipeline multiplies by 1/norm instead of dividing: fp64 division is ~25 instructions and
// the in-sampler scan is VALU-bound; every stage uses the same expression, so they agree bit for bit)
__global__ __launch_bounds__(QSMC_BLOCK) void k_chunk_sums(const double *__restrict__ w, int64_t n,
                                                           double inv_norm, double *__restrict__ sums) {
    __shared__ double lds[QSMC_WAVES_PER_BLOCK];
    const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
    double v[1] = {0.0};
#pragma unroll
    for (int u = 0; u < SCAN_CHUNK / QSMC_BLOCK; ++u) {
        const int64_t i = base + (int64_t)u * QSMC_BLOCK + threadIdx.x;
        if (i < n) v[0] += (w ? w[i] : 1.0) * inv_norm;
    }
    block_sum<1>(v, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = v[0];
}

__device__ __forceinline__ double wave_inclusive_max(double v, int lane) {
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const double t = __shfl_up(v, off, QSMC_WAVE);
        if (lane >= off) v = fmax(v, t);
    }
    return v;
}

// Exclusive scan of the chunk sums, in place, plus the grand total at sums[m]  (m + 1 outputs).
// Single 1024-thread workgroup; thread t owns a contiguous run of ceil(m / 1024) entries (serial,
// in registers), the 1024 run totals are scanned with wave shuffles + 16 wave totals.  Floating-point
// tree sums are not guaranteed monotone in the index, so the result goes through an exact prefix MAX
// (no rounding) in the same pass structure; k_chunk_scan relies on monotone offsets.
constexpr int SCAN_SUMS_THREADS = 1024;
constexpr int SCAN_SUMS_MAX_PER = 16;              // m <= 16384 chunks (N <= 6.7e7) in registers

// tiles != nullptr: chunk c's sum is (tiles[c tpc] + ... + tiles[c tpc + tpc - 1]) * inv_norm -- the per-tile, per-wave
// sums the last update kernel left behind (tpc = tiles per chunk x 4 waves) -- instead of sums[c] from k_chunk_sums.
struct TileSrc {
    const double *tiles;
    int tpc;
    int64_t n_tiles;
    double inv_norm;
};

__device__ __forceinline__ double chunk_sum_in(const double *__restrict__ sums, const TileSrc &ts, int64_t c) {
    if (!ts.tiles) return sums[c];
    if (ts.tpc == 8 && (c + 1) * 8 <= ts.n_tiles) {
        // the usual case (2 tiles x 4 waves): the chunk's eight parts are one 64-byte line -> two 32-byte loads,
        // summed in index order like the loop below
        const double4 a = *reinterpret_cast<const double4 *>(ts.tiles + c * 8);
        const double4 b = *reinterpret_cast<const double4 *>(ts.tiles + c * 8 + 4);
        return (((((((a.x + a.y) + a.z) + a.w) + b.x) + b.y) + b.z) + b.w) * ts.inv_norm;
    }
    double t = 0.0;
    for (int j = 0; j < ts.tpc; ++j) {
        const int64_t k = c * ts.tpc + j;
        if (k < ts.n_tiles) t += ts.tiles[k];
    }
    return t * ts.inv_norm;
}

// One workgroup of SCAN_SUMS_THREADS: exclusive, monotone prefix of the m chunk sums; sink(i, offsets[i]) for
// i = 0 .. m (offsets[m] = total).  The sums come from `sums` or, with ts.tiles, from the update kernel's tile sums.
template <int THREADS = SCAN_SUMS_THREADS, class Sink>
__device__ __forceinline__ void scan_sums_block(const double *sums, int64_t m, const TileSrc &ts, Sink sink) {
    __shared__ double wtot[THREADS / QSMC_WAVE];
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    const int per = (int)((m + THREADS - 1) / THREADS);
    const int64_t i0 = (int64_t)threadIdx.x * per;
    double v[SCAN_SUMS_MAX_PER];
    double run = 0.0;
    if (THREADS <= 256 && ts.tiles && ts.tpc == 8) {
        // (uniform; the reducing launch's prefix workgroup) 2 tiles x 4 waves per chunk: a chunk's eight parts are one
        // 64-byte line.  Thread t's RUN is `per` consecutive chunks, so loading it directly makes every wave instruction
        // touch 64 different lines, one memory latency after the other (7.8 us for 2442 chunks, against 4.3 us for the
        // reduction beside it).  Instead the chunk sums are formed lane-consecutive (chunk q THREADS + t: whole lines,
        // coalesced, every load of the thread in flight together), parked in LDS, and the runs are read from there:
        // the same sums added in the same order.  The update kernel zero-filled the last chunk's missing tiles.
        __shared__ double csum[THREADS <= 256 ? THREADS * SCAN_SUMS_MAX_PER : 1];
        double4 a[SCAN_SUMS_MAX_PER], b[SCAN_SUMS_MAX_PER];
#pragma unroll
        for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q) {
            if (q < per) {                                          // (uniform)
                const int64_t c = (int64_t)q * THREADS + threadIdx.x;
                const int64_t cc = c < m ? c : m - 1;
                a[q] = *reinterpret_cast<const double4 *>(ts.tiles + cc * 8);
                b[q] = *reinterpret_cast<const double4 *>(ts.tiles + cc * 8 + 4);
            }
        }
#pragma unroll
        for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q) {
            if (q < per)
                csum[q * THREADS + threadIdx.x] =
                    (((((((a[q].x + a[q].y) + a[q].z) + a[q].w) + b[q].x) + b[q].y) + b[q].z) + b[q].w) * ts.inv_norm;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q) {
            v[q] = (q < per && i0 + q < m) ? csum[i0 + q] : 0.0;
            run += v[q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q) {
            v[q] = (q < per && i0 + q < m) ? chunk_sum_in(sums, ts, i0 + q) : 0.0;
            run += v[q];
        }
    }
    // exclusive offset of this thread's run
    double inc = wave_inclusive_scan(run, lane);
    if (lane == QSMC_WAVE - 1) wtot[wave] = inc;
    __syncthreads();
    double off = inc - run;
    for (int wv = 0; wv < wave; ++wv) off += wtot[wv];
    double total = 0.0;
    for (int wv = 0; wv < THREADS / QSMC_WAVE; ++wv) total += wtot[wv];
    __syncthreads();
    // exclusive values of my entries, then make everything monotone with an exact prefix max
    double e[SCAN_SUMS_MAX_PER];
    double acc = off, mx = 0.0;
#pragma unroll
    for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q) {
        e[q] = acc;
        acc += v[q];
        mx = fmax(mx, e[q]);
        e[q] = mx;                                  // local running max (values are >= 0)
    }
    double wmx = wave_inclusive_max(mx, lane);
    if (lane == QSMC_WAVE - 1) wtot[wave] = wmx;
    __syncthreads();
    double before = __shfl_up(wmx, 1, QSMC_WAVE);   // max over earlier lanes of this wave
    if (lane == 0) before = 0.0;
    for (int wv = 0; wv < wave; ++wv) before = fmax(before, wtot[wv]);
#pragma unroll
    for (int q = 0; q < SCAN_SUMS_MAX_PER; ++q)
        if (q < per && i0 + q < m) sink(i0 + q, fmax(e[q], before));
    if (threadIdx.x == THREADS - 1) {
        double gmax = 0.0;
        for (int wv = 0; wv < THREADS / QSMC_WAVE; ++wv) gmax = fmax(gmax, wtot[wv]);
        sink(m, fmax(total, gmax));
    }
}

__global__ __launch_bounds__(SCAN_SUMS_THREADS) void k_scan_sums(double *sums, int64_t m,
                                                                 unsigned long long *__restrict__ zero2, TileSrc ts) {
    if (zero2 && threadIdx.x < 2) zero2[threadIdx.x] = 0ull;      // the resampler's failed / retry counters (was a memset launch)
    scan_sums_block(sums, m, ts, [&](int64_t i, double v) { sums[i] = v; });
}

// The reducing launch of an update that left tile sums, with a second workgroup: while workgroup 0 reduces the
// partials (and decides the resample gate), workgroup 1 forms the monotone prefix of the chunk sums from the tile
// sums -- UNNORMALISED: the normaliser is what workgroup 0 is computing -- so that k_bucket_counts, should a resample
// be due, starts from 20 KB of ready offsets (one multiply each) instead of scanning 156 KB of tile sums itself
// (10 of its 28 us at N = 1e7, in every one of its 16 workgroups).  On the other steps the prefix costs nothing on
// the critical path: it runs beside the reduction, inside the host's round trip.
constexpr int TILE_PREFIX_MAX_CHUNKS = QSMC_BLOCK * SCAN_SUMS_MAX_PER;        // 4096 chunks: N <= 1.67e7
template <int NS>
__global__ __launch_bounds__(QSMC_BLOCK) void k_reduce_partials_scan(int nblocks, ReduceOut ro) {
    if (blockIdx.x == 0) {
        reduce_partials_body<NS>(nblocks, ro);
        return;
    }
    const TileSrc ts{ro.tile_sums, ro.tp_tpc, (int64_t)ro.tp_ntiles, 1.0};
    scan_sums_block<QSMC_BLOCK>(nullptr, (int64_t)ro.tp_chunks, ts, [&](int64_t i, double v) { ro.tile_prefix[i] = v; });
}

// Fallback for m > 16384 chunk sums (N > 6.7e7): same contract, 256-wide slabs with a carry.
__global__ __launch_bounds__(QSMC_BLOCK) void k_scan_sums_big(double *__restrict__ sums, int64_t m,
                                                              unsigned long long *__restrict__ zero2, TileSrc ts) {
    __shared__ double wave_tot[QSMC_WAVES_PER_BLOCK];
    if (zero2 && threadIdx.x < 2) zero2[threadIdx.x] = 0ull;
    __shared__ double carry_s;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    if (threadIdx.x == 0) carry_s = 0.0;
    __syncthreads();
    for (int64_t base = 0; base < m; base += QSMC_BLOCK) {
        const int64_t i = base + threadIdx.x;
        const double v = i < m ? chunk_sum_in(sums, ts, i) : 0.0;
        const double inc = wave_inclusive_scan(v, lane);
        double excl = __shfl_up(inc, 1, QSMC_WAVE);
        if (lane == 0) excl = 0.0;
        if (lane == QSMC_WAVE - 1) wave_tot[wave] = inc;
        __syncthreads();
        double off = carry_s;
        for (int wv = 0; wv < wave; ++wv) off += wave_tot[wv];
        if (i < m) sums[i] = off + excl;
        __syncthreads();
        if (threadIdx.x == QSMC_BLOCK - 1) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) sums[m] = carry_s;
    __syncthreads();
    if (threadIdx.x == 0) carry_s = 0.0;
    __syncthreads();
    for (int64_t base = 0; base <= m; base += QSMC_BLOCK) {      // exact prefix max over sums[0..m]
        const int64_t i = base + threadIdx.x;
        const double v = i <= m ? sums[i] : 0.0;
        const double mx = wave_inclusive_max(v, lane);
        if (lane == QSMC_WAVE - 1) wave_tot[wave] = mx;
        __syncthreads();
        double run = carry_s;
        for (int wv = 0; wv < wave; ++wv) run = fmax(run, wave_tot[wv]);
        const double out = fmax(mx, run);
        if (i <= m) sums[i] = out;
        __syncthreads();
        if (threadIdx.x == QSMC_BLOCK - 1) carry_s = out;
        __syncthreads();
    }
}

// Per-chunk scan.  offsets[] has chunks + 1 monotone entries (exclusive offsets + total).
// Lane l of wave v owns the 8 consecutive particles [512 v + 8 l, +8): a serial running sum in
// registers (monotone by construction), one wave scan of the lane totals, 8 wave totals through LDS.
// Every value is clamped into its wave's [lo, hi] offset window and the lanes' last values go through
// an exact prefix max, so the CDF is non-decreasing everywhere (searchsorted on it is well defined)
// while differing from the sequential np.cumsum only by rounding.  The last entry of a wave's
// 512-particle segment is DEFINED as the window top hi, and the chunk's last entry as offsets[c + 1]
// (equal in exact arithmetic), so chunk edges and CDF entries are one and the same numbers whether or
// not the CDF is ever written to HBM, and each lane knows its predecessor's value without a barrier.
// The first 512 threads scan chunk c; every thread of the workgroup must call (one barrier inside).
// store(j, value, prev) receives the chunk-local index, the entry and the entry before it (the chunk's
// lower edge for j = 0); calls are made wave-uniformly (`live` = the entry exists).
template <class Store>
__device__ __forceinline__ void chunk_scan_block(const double *__restrict__ w, int64_t n, double inv_norm,
                                                 const double *__restrict__ offsets, int64_t c,
                                                 double *wave_tot, Store store) {
    const bool act = threadIdx.x < SCAN_THREADS;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = (threadIdx.x / QSMC_WAVE) & (SCAN_WAVES - 1);
    const int j0 = wave * SCAN_WAVE_CHUNK + lane * SCAN_PER_LANE;          // chunk-local index of v[0]
    const int64_t i0 = c * SCAN_CHUNK + j0;
    double v[SCAN_PER_LANE];
    double excl = 0.0;
    if (act) {
        if (!w) {
#pragma unroll
            for (int k = 0; k < SCAN_PER_LANE; ++k) v[k] = i0 + k < n ? inv_norm : 0.0;
        } else if (i0 + SCAN_PER_LANE <= n && ((uintptr_t)w & 31) == 0) {
#pragma unroll
            for (int k = 0; k < SCAN_PER_LANE; k += 4) {
                const double4 t = *reinterpret_cast<const double4 *>(w + i0 + k);
                v[k] = t.x * inv_norm; v[k + 1] = t.y * inv_norm; v[k + 2] = t.z * inv_norm; v[k + 3] = t.w * inv_norm;
            }
        } else {
#pragma unroll
            for (int k = 0; k < SCAN_PER_LANE; ++k) v[k] = i0 + k < n ? w[i0 + k] * inv_norm : 0.0;
        }
#pragma unroll
        for (int k = 1; k < SCAN_PER_LANE; ++k) v[k] += v[k - 1];
        const double inc = wave_inclusive_scan(v[SCAN_PER_LANE - 1], lane);
        excl = __shfl_up(inc, 1, QSMC_WAVE);
        if (lane == 0) excl = 0.0;
        if (lane == QSMC_WAVE - 1) wave_tot[wave] = inc;
    }
    __syncthreads();
    if (!act) return;
    const double blo = c <= 0 ? 0.0 : offsets[c], bhi = offsets[c + 1];
    const int len = (int)((n - c * SCAN_CHUNK) < SCAN_CHUNK ? (n - c * SCAN_CHUNK) : SCAN_CHUNK);
    double lo = blo;
    for (int wv = 0; wv < wave; ++wv) lo = fmin(lo + wave_tot[wv], bhi);
    const double hi = (wave == SCAN_WAVES - 1) ? bhi : fmin(lo + wave_tot[wave], bhi);
#pragma unroll
    for (int k = 0; k < SCAN_PER_LANE; ++k) v[k] = fmin(fmax(lo + (excl + v[k]), lo), hi);
    const double m = wave_inclusive_max(v[SCAN_PER_LANE - 1], lane);
    double prev = __shfl_up(m, 1, QSMC_WAVE);
    if (lane == 0) prev = lo;                      // == the previous wave's (forced) last entry, or the chunk's lower edge
#pragma unroll
    for (int k = 0; k < SCAN_PER_LANE; ++k) {
        const int j = j0 + k;
        double a = fmax(v[k], prev);
        if (j == len - 1) a = bhi;
        else if (lane == QSMC_WAVE - 1 && k == SCAN_PER_LANE - 1) a = hi;
        store(j, a, prev, j < len);
        prev = a;
    }
}

struct StoreGlobal {
    double *cdf;                                   // + chunk base
    __device__ __forceinline__ void operator()(int j, double v, double, bool live) const {
        if (live) cdf[j] = v;
    }
};

// Materialise the CDF.  gate != nullptr: do nothing unless *gate > 0 (the bucketed resampler only
// needs the global CDF when some particle has to redraw a global ancestor).
__global__ __launch_bounds__(SCAN_THREADS) void k_chunk_scan(const double *__restrict__ w, int64_t n,
                                                             double inv_norm, const double *__restrict__ offsets,
                                                             double *__restrict__ cdf,
                                                             const unsigned long long *__restrict__ gate) {
    __shared__ double wave_tot[SCAN_WAVES];
    if (gate && *gate == 0ull) return;
    chunk_scan_block(w, n, inv_norm, offsets, (int64_t)blockIdx.x, wave_tot,
                     StoreGlobal{cdf + (int64_t)blockIdx.x * SCAN_CHUNK});
}

