// kernels/resample.hpp -- Liu-West resampling: legacy-RNG pieces, the direct device-RNG sampler and the bucketed multinomial resampler.
// Part of the single translation unit qsmc_kernels.hip (included there, in this order; not a stand-alone header).
#pragma once

// =============================================================================================
// Liu-West pieces
// =============================================================================================
// upper bound: number of entries <= u, clamped to n - 1
__device__ __forceinline__ int64_t search_right(const double *__restrict__ cdf, int64_t n, double u) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo < n - 1 ? lo : n - 1;
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_ancestors(const double *__restrict__ cdf, int64_t n_in,
                                                          const double *__restrict__ u, int64_t n_out,
                                                          int64_t *__restrict__ js) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out;
         i += (int64_t)gridDim.x * QSMC_BLOCK)
        js[i] = search_right(cdf, n_in, u[i]);
}

struct LWArgs {
    double a;
    double mean[QSMC_MAX_D];
    double S[QSMC_MAX_D * QSMC_MAX_D];   // row-major d x d (already times h)
};

__global__ __launch_bounds__(QSMC_BLOCK) void k_centres(const double *__restrict__ x_in, int64_t ldx_in,
                                                        int d, const int64_t *__restrict__ js,
                                                        int64_t n_out, double a, LWArgs lw,
                                                        double *__restrict__ mus, int64_t ld_mus) {
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t j = js[i];
        for (int m = 0; m < d; ++m)
            mus[m * ld_mus + i] = a * x_in[m * ldx_in + j] + (1.0 - a) * lw.mean[m];   // :325
    }
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_perturb(int kind, int d, double min_freq, int postselect,
                                                        const double *__restrict__ mus, int64_t ld_mus,
                                                        const int64_t *__restrict__ idxs, int64_t k,
                                                        int centre_by_idx, LWArgs lw,
                                                        const double *__restrict__ z, int64_t ldz,
                                                        double *__restrict__ x_out, int64_t ldx_out,
                                                        uint8_t *__restrict__ valid) {
    for (int64_t r = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; r < k;
         r += (int64_t)gridDim.x * QSMC_BLOCK) {
        const int64_t dst = idxs ? idxs[r] : r;
        const int64_t c = centre_by_idx ? dst : r;
        double p[QSMC_MAX_D];
        for (int m = 0; m < d; ++m) {
            double s = 0.0;                     // (S @ z)[m, r], summed in column order like np.dot
            for (int q = 0; q < d; ++q) s += lw.S[m * d + q] * z[q * ldz + r];
            p[m] = mus[m * ld_mus + c] + s;
            x_out[m * ldx_out + dst] = p[m];
        }
        valid[r] = (!postselect || model_valid(kind, p, min_freq)) ? 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Output placement.  Single GPU: slot o -> column o of the SoA cloud.  Sharded: this
// rank produces the finished particles for EVERY destination rank and they leave by one
// all_to_all, so rows must be grouped by destination, AoS.  The bucketed sampler emits slots sorted
// by ancestor chunk; dealing them to destinations round-robin (skipping a destination once its
// quota is full -- exact quotas, closed form below) gives every destination an even, stratified
// share of all chunks, so the shards stay statistically exchangeable.
// ---------------------------------------------------------------------------------------------
#define QSMC_MAX_DEST 16
struct OutPlace {
    int n_dest;                            // 0: identity placement
    int64_t ld_m, ld_s;                    // element (m, row) lives at x_out[m * ld_m + row * ld_s]
    int order[QSMC_MAX_DEST];              // destination ids by ascending quota
    int64_t quota[QSMC_MAX_DEST];          // ascending quotas c_(0) <= ... <= c_(G-1)
    int64_t seg_start[QSMC_MAX_DEST + 1];  // first slot of dealing segment s (rounds c_(s-1) .. c_(s)-1)
    int64_t dest_base[QSMC_MAX_DEST];      // first row of destination r
};

__device__ __forceinline__ int64_t place_row(const OutPlace &pl, int64_t o) {
    if (pl.n_dest == 0) return o;
    int s = 0;
    while (s + 1 < pl.n_dest && o >= pl.seg_start[s + 1]) ++s;
    const int active = pl.n_dest - s;
    const int64_t rel = o - pl.seg_start[s];
    const int64_t round = (s ? pl.quota[s - 1] : 0) + rel / active;
    const int dest = pl.order[s + (int)(rel % active)];
    return pl.dest_base[dest] + round;
}

// One-launch device-RNG resample.
__global__ __launch_bounds__(QSMC_BLOCK) void k_resample_philox(
    int kind, int d, double min_freq, int postselect, const double *__restrict__ x_in, int64_t ldx_in,
    int64_t n_in, const double *__restrict__ cdf, LWArgs lw, int64_t n_out, uint32_t k0, uint32_t k1,
    uint32_t epoch, int maxiter, double *__restrict__ x_out, OutPlace pl,
    unsigned long long *__restrict__ n_failed) {
    unsigned long long failed = 0;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n_out;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[QSMC_MAX_D];
        bool ok = false;
        for (int round = 0; round < maxiter && !ok; ++round) {
            PhiloxStream rng{(uint64_t)i, (epoch << 16) | (uint32_t)round, k0, k1};
            double u, unused;
            rng.uniforms(0, u, unused);
            const int64_t j = search_right(cdf, n_in, u);
            double zz[QSMC_MAX_D];
            for (int q = 0; q < d; q += 2) {
                double z0, z1;
                rng.normals(1 + (q >> 1), z0, z1);
                zz[q] = z0;
                if (q + 1 < d) zz[q + 1] = z1;
            }
            for (int m = 0; m < d; ++m) {
                double s = 0.0;
                for (int q = 0; q < d; ++q) s += lw.S[m * d + q] * zz[q];
                p[m] = (lw.a * x_in[m * ldx_in + j] + (1.0 - lw.a) * lw.mean[m]) + s;
            }
            ok = !postselect || model_valid(kind, p, min_freq);
        }
        const int64_t row = place_row(pl, i);
        for (int m = 0; m < d; ++m) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
        if (!ok) ++failed;
    }
    if (failed) atomicAdd(n_failed, failed);
}


// =============================================================================================
// Bucketed multinomial resampling (device RNG).
//
// The direct kernel above does one 23-level binary search of the 80 MB CDF per output particle:
// ~1.2e8 scattered line requests at N = 1e7 (measured 1.4 ms, 83 % of GPU time in round-1
// profile a).  Output particles are exchangeable, so instead:
//   A  chunk counts      how many outputs descend from each CDF CHUNK (4096 source particles): exact
//                        Multinomial(N; W_chunk) counts.  k_bucket_counts: independent Poisson
//                        draws per chunk plus a short categorical top-up (see "Poissonisation" below);
//                        k_bucket_count / k_bucket_reduce (QSMC_COUNT_BY_DRAWS=1, the first implementation):
//                        every output draws u_i and is binned against the chunk edges in LDS;
//   B  (k_bucket_counts, last step) / k_bucket_plan   exclusive scans: first output slot of each
//                        chunk and a work list that splits heavy chunks into <= BUCKET_CAP outputs;
//   C  k_bucket_sample   one workgroup per work item scans ITS chunk of the weights into LDS (32 KB of CDF),
//                        draws the within-chunk position from an independent Philox word (given
//                        the counts, positions are i.i.d. uniform inside the chunk -- exact),
//                        searches in LDS, gathers x from the chunk's 32 KB window, kicks, checks
//                        validity and writes its outputs to consecutive slots.
// HBM traffic becomes streaming (read w + x once, write x' once); all scattered probes hit LDS.
// A postselection retry needs a fresh GLOBAL ancestor: that rare path falls back to the global
// search (same semantics as k_resample_philox: redraw ancestor and kick).
// =============================================================================================
constexpr int BUCKET_CHUNK = SCAN_CHUNK;            // 4096 source particles per bucket
constexpr int BUCKET_CAP = 2 * BUCKET_CHUNK;        // outputs per work item, at most (the host picks `cap` <= this: bucket_cap())
constexpr int BUCKET_MAX_CHUNKS = 8192;             // skewed edges (68 KB) + counters (32 KB) + guide (16 KB) of LDS
constexpr int BUCKET_COUNT_BLOCKS = 256;            // one resident workgroup per CU
constexpr int BUCKET_COUNT_THREADS = 1024;

// Lower CDF edge of chunk c == upper edge of chunk c-1 == offsets[c] (k_scan_sums output; the chunk
// scan defines the last CDF entry of every chunk as exactly this number).
__device__ __forceinline__ double chunk_edge(const double *__restrict__ offsets, int64_t c) {
    return c <= 0 ? 0.0 : offsets[c];
}

// number of entries of a[0..m) that are <= u   (a in LDS or global)
__device__ __forceinline__ int upper_bound_i32(const double *a, int m, double u) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// LDS index skew: binary-search midpoints of a power-of-two table are multiples of 2048, 1024, ...
// elements, i.e. ONE bank for every lane (measured: 88 % of the sample kernel's LDS cycles were bank
// conflicts).  j + (j >> 5) + (j >> 10) sends those strides to distinct banks.
__device__ __forceinline__ int lds_skew(int j) { return j + (j >> 5) + (j >> 10); }
constexpr int BUCKET_CHUNK_LDS = BUCKET_CHUNK + (BUCKET_CHUNK >> 5) + (BUCKET_CHUNK >> 10) + 4;

// number of entries of the SKEWED LDS table a[skew(0..m)) that are <= u
__device__ __forceinline__ int upper_bound_skew(const double *a, int m, double u) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[lds_skew(mid)] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------------------
// Guide table: a 12-probe binary search of a 4096-entry LDS table costs ~12 instructions per probe.
// A uniform grid of cells over the table's value range, G[k] = #{entries whose cell < k}, brackets the
// answer for a query in cell k inside [G[k], G[k+1]] -- exactly, because guide_cell() is monotone and is
// applied identically to the entries and to the query -- typically 1-2 entries -> ~1 probe.
// k_bucket_count builds its table (cells over the chunk edges) with an LDS histogram + scan
// (build_guide); k_bucket_sample fills its table while it stores the scanned CDF (StoreLdsGuide).
// Exactness is unaffected either way: the answer always comes from comparing the entries with u.
// ---------------------------------------------------------------------------------------------
constexpr int GUIDE_BINS = 4096;              // k_bucket_count: cells over [0, 1) for the chunk edges
constexpr int SGUIDE_BINS = 2048;             // k_bucket_sample: cells over one chunk's 4096 CDF entries

template <int BINS>
__device__ __forceinline__ int guide_cell(double v, double lo, double scale) {
    const double t = (v - lo) * scale;
    int k = t > 0.0 ? (t < (double)(BINS - 1) ? (int)t : BINS - 1) : 0;
    return k;
}

// a: skewed LDS table of m non-decreasing values; G: int[GUIDE_BINS + 1] LDS; wtot: int[32] LDS.
template <int BT>
__device__ __forceinline__ void build_guide(const double *a, int m, double lo, double scale, int *G, int *wtot) {
    constexpr int PER = GUIDE_BINS / BT;
    static_assert(GUIDE_BINS % BT == 0, "GUIDE_BINS must be a multiple of the workgroup size");
    for (int k = threadIdx.x; k <= GUIDE_BINS; k += BT) G[k] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < m; j += BT) atomicAdd(&G[guide_cell<GUIDE_BINS>(a[lds_skew(j)], lo, scale) + 1], 1);
    __syncthreads();
    // inclusive scan of G[1..GUIDE_BINS]: thread owns PER consecutive cells
    int loc[PER];
    int run = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        run += G[1 + threadIdx.x * PER + q];
        loc[q] = run;
    }
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    int inc = run;
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const int t = __shfl_up(inc, off, QSMC_WAVE);
        if (lane >= off) inc += t;
    }
    if (lane == QSMC_WAVE - 1) wtot[wave] = inc;
    __syncthreads();
    int off0 = inc - run;
    for (int wv = 0; wv < wave; ++wv) off0 += wtot[wv];
#pragma unroll
    for (int q = 0; q < PER; ++q) G[1 + threadIdx.x * PER + q] = off0 + loc[q];
    __syncthreads();
}

// number of entries of the skewed table a[0..m) that are <= u, u lying in guide cell k
// guide_cell is monotone and is applied identically to the entries and to u, so entries in cells
// below k are <= u and entries in cells above k are > u: the answer lies in [G[k], G[k + 1]] exactly.
template <class GT>
__device__ __forceinline__ int guided_upper_bound(const double *a, int m, const GT *G, int k, double u) {
    int lo = G[k];
    int hi = G[k + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[lds_skew(mid)] <= u) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Philox stream layout of the bucketed resampler (round 0), two outputs per Philox block:
//   slot 0: block (c | t << 32)                   attempt t of chunk c's Poisson count (k_bucket_counts)
//           [QSMC_COUNT_BY_DRAWS: block (i >> 1), word (i & 1) = chunk draw of output i (k_bucket_count)]
//   slot 3: block (j >> 1), word (j & 1)          top-up draw j;  slot 4: block (i), word 0: removal i
//   slot 1: block (o >> 1), word (o & 1)          within-chunk position of slot o (k_bucket_sample)
//   slot 2: block (n >> 1), Box-Muller comp (n&1) n = o * d + q, q-th normal of slot o
// retries (round r >= 1) are per output: block (o, r, 0).u0 = global ancestor, (o, r, 1 + q/2) normals.
__global__ __launch_bounds__(BUCKET_COUNT_THREADS) void k_bucket_count(
    const double *__restrict__ offsets, int chunks, int64_t n_out, uint32_t k0, uint32_t k1,
    uint32_t epoch, unsigned int *__restrict__ hist /* [gridDim.x][chunks] */) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double *edges = reinterpret_cast<double *>(smem);                       // skewed: upper edge of chunk c
    const int edges_len = lds_skew(chunks) + 4;
    unsigned int *cnt = reinterpret_cast<unsigned int *>(edges + edges_len);
    int *G = reinterpret_cast<int *>(cnt + chunks);
    int *wtot = G + GUIDE_BINS + 1;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) {
        edges[lds_skew(c)] = chunk_edge(offsets, (int64_t)c + 1);
        cnt[c] = 0u;
    }
    __syncthreads();
    build_guide<BUCKET_COUNT_THREADS>(edges, chunks, 0.0, (double)GUIDE_BINS, G, wtot);
    const int64_t n_pairs = (n_out + 1) >> 1;
    for (int64_t pr = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pr < n_pairs;
         pr += (int64_t)gridDim.x * blockDim.x) {
        PhiloxStream rng{(uint64_t)pr, (epoch << 16), k0, k1};
        double u[2];
        rng.uniforms(0, u[0], u[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (2 * pr + e < n_out) {
                // #edges <= u == chunk index; u in [0, 1) so its guide cell is exactly floor(u * 4096)
                int c = guided_upper_bound(edges, chunks, G, (int)(u[e] * (double)GUIDE_BINS), u[e]);
                if (c > chunks - 1) c = chunks - 1;              // u beyond cdf[n-1] (rounding): Q2 clamp
                atomicAdd(&cnt[c], 1u);
            }
        }
    }
    __syncthreads();
    unsigned int *row = hist + (size_t)blockIdx.x * chunks;
    for (int c = threadIdx.x; c < chunks; c += blockDim.x) row[c] = cnt[c];
}

// ---------------------------------------------------------------------------------------------
// Chunk counts without drawing one uniform per output (k_bucket_count + k_bucket_reduce: 32 us at N = 1e7).
// Poissonisation: if T ~ Poisson(lambda) items are dealt to the chunks with probabilities p_c, the chunk
// counts are INDEPENDENT Poisson(lambda p_c) -- one draw per chunk, all in parallel, no tree and no depth --
// and given T they are Multinomial(T; p).  With lambda = n_out - kappa sqrt(n_out) (kappa = 5) T falls short of
// n_out by ~kappa sqrt(n_out) outputs, which are added as ordinary categorical draws (one uniform each,
// searched against the chunk edges: 1.6e4 draws instead of 1e7); Multinomial(T) + Multinomial(n_out - T) =
// Multinomial(n_out), exactly the law k_bucket_count samples.  Should T exceed n_out (probability 3e-7 per
// resample) the surplus is taken away again by removing T - n_out of the dealt items uniformly at random,
// which leaves an i.i.d. sample of size n_out: exact as well.
//
// poisson_draw: X ~ Poisson(mu), exact.  mu < 10: sequential search of the cdf from X = 0; otherwise PTRS
// (W. Hoermann, "The transformed rejection method for generating Poisson random variables", Insurance:
// Mathematics and Economics 12 (1993) 39): a squeeze accepts ~87 % of the proposals after one division and a
// floor.  Attempt t of chunk c takes its uniforms from Philox block (c | t << 32, round 0, slot 0) and the
// FIRST accepted attempt is the draw; G adjacent lanes evaluate attempts t0 .. t0 + G - 1 of one chunk at
// once and the lowest accepted one is taken (ballot + shuffle) -- the value a sequential loop returns, which
// is how the oracle's NumPy twin (oracle/philox.py: poisson_draw) computes it.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double stirling_tail(double k) {         // ln k! - [(k + 1/2) ln(k + 1) - (k + 1) + ln(2 pi) / 2]
    static constexpr double small[10] = {
        0.08106146679532726,  0.0413406959554093,   0.02767792568499834,  0.020790672103765093, 0.016644691189821193,
        0.013876128823070748, 0.01189670994589177,  0.010411265261972096, 0.009255462182712733, 0.00833056343336287};
    if (k < 10.0) return small[(int)k];
    const double rx = 1.0 / (k + 1.0), r2 = rx * rx;
    return (1.0 / 12.0 - (1.0 / 360.0 - (1.0 / 1260.0 - (1.0 / 1680.0 - 1.0 / 1188.0 * r2) * r2) * r2) * r2) * rx;
}

// Called by whole waves.  Lanes [gbase, gbase + G) of a wave form the group of one chunk (same mu, node, active);
// G is a power of two <= 64.  Returns the draw to every lane of the group.
__device__ unsigned int poisson_draw(bool active, double mu, uint32_t node, uint32_t epoch_round, uint32_t k0,
                                     uint32_t k1, int G, int gbase) {
    const bool need = active && mu > 0.0;
    double y = 0.0;
    const bool by_search = need && mu < 10.0;
    if (by_search) {                                             // every lane of the group: same inputs, same value
        PhiloxStream rng{(uint64_t)node, epoch_round, k0, k1};
        double U, unused;
        rng.uniforms(0, U, unused);
        double pk = exp(-mu), cdf = pk, X = 0.0;
        while (U > cdf && X < 200.0) {
            X += 1.0;
            pk = pk * mu / X;
            cdf += pk;
        }
        y = X;
    }
    bool pending = need && !by_search;
    const double smu = sqrt(mu), lmu = log(mu);
    const double b = 0.931 + 2.53 * smu, a = -0.059 + 0.02483 * b;
    const double linva = log(1.1239 + 1.1328 / (b - 3.4)), vr = 0.9277 - 3.6224 / (b - 2.0);
    const int gl = (int)(threadIdx.x & (QSMC_WAVE - 1)) - gbase;
    const unsigned long long gmask = G >= 64 ? ~0ull : ((1ull << G) - 1ull);
    for (uint32_t t0 = 0; t0 < 4096u; t0 += (uint32_t)G) {
        if (__ballot(pending) == 0ull) break;                    // wave-uniform
        bool acc = false;
        double kk = 0.0;
        if (pending) {
            PhiloxStream rng{(uint64_t)node | ((uint64_t)(t0 + (uint32_t)gl) << 32), epoch_round, k0, k1};
            double U, V;
            rng.uniforms(0, U, V);
            const double u = U - 0.5, us = 0.5 - fabs(u);
            kk = floor((2.0 * a / us + b) * u + mu + 0.43);
            if (us >= 0.07 && V <= vr) acc = true;               // the squeeze
            else if (kk >= 0.0 && !(us < 0.013 && V > us)) {
                const double lhs = log(V) + linva - log(a / (us * us) + b);
                const double lgk = (kk + 0.5) * log(kk + 1.0) - (kk + 1.0) + 0.91893853320467274178 + stirling_tail(kk);
                acc = lhs <= -mu + kk * lmu - lgk;
            }
        }
        const unsigned long long grp = (__ballot(acc) >> gbase) & gmask;
        const int src = gbase + (grp ? __builtin_ctzll(grp) : 0);
        const double first = __shfl(kk, src, QSMC_WAVE);
        if (pending && grp) {
            y = first;
            pending = false;
        }
    }
    return need ? (unsigned int)y : 0u;
}

// single workgroup (1024 threads): slot_off[c] = exclusive scan of counts; item_off[c] = exclusive scan of
// ceil(counts / BUCKET_CAP); slot_off[chunks] = n_out, item_off[chunks] = #work items.  counts: global or LDS.
__device__ __forceinline__ void bucket_plan_block(const unsigned int *counts, int chunks, int cap,
                                                  long long *__restrict__ slot_off, int *__restrict__ item_off,
                                                  int *__restrict__ item_chunk) {
    __shared__ long long wtot_s[1024 / QSMC_WAVE];
    __shared__ int wtot_i[1024 / QSMC_WAVE];
    const int per = (chunks + 1023) / 1024;
    const int c0 = threadIdx.x * per, c1 = min(chunks, c0 + per);
    long long s = 0;
    int it = 0;
    for (int c = c0; c < c1; ++c) {
        s += counts[c];
        it += (int)((counts[c] + (unsigned)cap - 1u) / (unsigned)cap);
    }
    // exclusive scan of the 1024 per-thread totals (integers: exact): shuffles inside a wave, 16 wave totals in LDS
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    long long inc_s = s;
    int inc_i = it;
#pragma unroll
    for (int off = 1; off < QSMC_WAVE; off <<= 1) {
        const long long ts = __shfl_up(inc_s, off, QSMC_WAVE);
        const int ti = __shfl_up(inc_i, off, QSMC_WAVE);
        if (lane >= off) {
            inc_s += ts;
            inc_i += ti;
        }
    }
    if (lane == QSMC_WAVE - 1) {
        wtot_s[wave] = inc_s;
        wtot_i[wave] = inc_i;
    }
    __syncthreads();
    long long so = inc_s - s;
    int io = inc_i - it;
    for (int wv = 0; wv < wave; ++wv) {
        so += wtot_s[wv];
        io += wtot_i[wv];
    }
    for (int c = c0; c < c1; ++c) {
        slot_off[c] = so;
        item_off[c] = io;
        const int items = (int)((counts[c] + (unsigned)cap - 1u) / (unsigned)cap);
        for (int k = 0; k < items; ++k) item_chunk[io + k] = c;     // work item -> chunk map
        so += counts[c];
        io += items;
    }
    if (threadIdx.x == 1023) {                                   // (after the loop: so / io have run through its chunks)
        slot_off[chunks] = so;
        item_off[chunks] = io;
    }
}

// Barrier across the workgroups of ONE launch whose grid is small enough to be resident at once (16 here).  The
// arrival counter only ever grows: the host hands every launch the value all workgroups will have brought it to
// at each of its barriers, so nothing is reset.  No cache maintenance: the XCDs' L2s are not coherent with each
// other inside a launch, and a release / acquire fence pair at agent scope (L2 write-back + invalidate) measured
// ~5 us per barrier -- instead every word that crosses workgroups is written and read with agent-scope atomics
// (which go to the coherence point), and __syncthreads() has waited for this workgroup's own before the
// arrival is posted.  A bounded spin (~1 s): a launch that cannot become resident aborts (the next HIP call
// reports it) rather than hanging the queue or carrying on with half the data.
__device__ __forceinline__ void grid_barrier(unsigned long long *bar, unsigned long long target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 20)) {                          // cannot happen with a resident grid: fail loudly
                __hip_atomic_fetch_add(bar + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_trap();
            }
        }
    }
    __syncthreads();
}
// A barrier with the cache maintenance, for bulk data written with plain stores (the redraw kernel's CDF): release
// (L2 write-back) before the arrival, acquire (invalidate) after the wait; ~5 us, on a path most resamples skip.
// Self-resetting (bar[0] arrivals, bar[1] departures: the last workgroup to leave clears both -- nobody can still
// be waiting then), so a launch that never reaches the barrier touches nothing.
__device__ __forceinline__ void grid_barrier_fenced(unsigned long long *bar) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)gridDim.x) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 20)) __builtin_trap();          // cannot happen with a resident grid: fail loudly
        }
        if (__hip_atomic_fetch_add(bar + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned long long)gridDim.x - 1ull) {
            __hip_atomic_store(bar + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(bar, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __threadfence();                 // acquire for the whole workgroup: the CU's L1 and the XCD's L2 are shared
    }
    __syncthreads();
}
__device__ __forceinline__ void put_shared(unsigned int *p, unsigned int v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int get_shared(const unsigned int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The chunk counts in one launch (each of the three steps alone is a ~5 us launch: the floor of a dependent
// kernel on this part):
//   0  (if the update kernel left tile sums) the chunk edges: see below;
//   1  counts[c] ~ Poisson(lambda mass_c / total), four lanes per chunk, chunks dealt to the workgroups;
//   2  every workgroup sums the counts to T (a few thousand integers), and takes its share of the n_out - T
//      categorical top-up draws against the chunk edges in LDS (1.6e4 draws on one CU were 25 us; spread over
//      16 they are 2), collected in an LDS histogram and added to extra[];  top-up draw j takes word (j & 1) of
//      Philox block (j >> 1, round 0, slot 3);
//   3  workgroup 0: counts += extra; should the Poisson total have overshot, thread 0 removes the surplus item
//      by item (removal i: word 0 of block (i, round 0, slot 4)); then the plan.
constexpr int POISSON_G = 4;
constexpr int BUCKET_COUNTS_BLOCKS = 16, BUCKET_COUNTS_THREADS = 1024;     // (8 workgroups: +4 us; 32: no gain)
__global__ __launch_bounds__(BUCKET_COUNTS_THREADS) void k_bucket_counts(
    double *offsets, TileSrc ts, const double *__restrict__ tile_prefix, unsigned long long *__restrict__ zero2, int chunks, int64_t n_out, double lambda,
    uint32_t k0, uint32_t k1, uint32_t epoch, unsigned int *counts, unsigned int *extra,
    long long *__restrict__ slot_off, int *__restrict__ item_off, int *__restrict__ item_chunk,
    unsigned long long *bar, unsigned long long bar_base, int cap, const int *__restrict__ gate,
    const double *__restrict__ norm_dev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (gate) {
        // queued speculatively behind an update (qsmc_lw_arm_prefix): the reducing kernel decided whether a resample
        // is due.  Not due: leave at once, having made this launch's two barrier arrivals (the counter only grows).
        if (*gate == 0) {
            if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 2ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        ts.inv_norm = 1.0 / *norm_dev;             // sum w' of that update, as the host will pass it to the sampler
    }
    double *edges = reinterpret_cast<double *>(smem);
    unsigned int *hist = reinterpret_cast<unsigned int *>(edges + lds_skew(chunks) + 4);   // this workgroup's draws per chunk
    __shared__ unsigned long long total_s;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    // ---- 0: the chunk edges.  Given the update kernel's tile sums (ts.tiles), EVERY workgroup forms the monotone
    // prefix of the chunk sums itself, straight into its LDS -- the same code on the same numbers in the same
    // order, so all agree bit for bit, and the separate one-workgroup scan launch (9 us) is gone; workgroup 0
    // also stores offsets[] for the sampler and clears the failed / retry counters.  Otherwise offsets[] is ready.
    static_assert(BUCKET_COUNTS_THREADS == SCAN_SUMS_THREADS, "scan_sums_block runs on this workgroup");
    if (ts.tiles && tile_prefix) {
        // (round 3) the prefix of the unnormalised chunk sums is ready -- the update's reducing launch formed it beside
        // the reduction (k_reduce_partials_scan): one multiply per edge; scaling by a positive number keeps it monotone
        const bool writer = blockIdx.x == 0;
        if (writer && threadIdx.x < 2) zero2[threadIdx.x] = 0ull;
        for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS) {
            const double v = tile_prefix[c + 1] * ts.inv_norm;
            edges[lds_skew(c)] = v;
            if (writer) offsets[c + 1] = v;
        }
        if (writer && threadIdx.x == 0) offsets[0] = 0.0;
    } else if (ts.tiles) {
        const bool writer = blockIdx.x == 0;
        if (writer && threadIdx.x < 2) zero2[threadIdx.x] = 0ull;
        scan_sums_block(nullptr, (int64_t)chunks, ts, [&](int64_t i, double v) {
            if (i > 0) edges[lds_skew((int)i - 1)] = v;             // upper edge of chunk i - 1
            if (writer) offsets[i] = v;
        });
    } else {
        for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS) edges[lds_skew(c)] = offsets[c + 1];
    }
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS) hist[c] = 0u;
    if (threadIdx.x == 0) total_s = 0ull;
    __syncthreads();
    // ---- 1: Poisson counts ----
    const double total = edges[lds_skew(chunks - 1)];
    constexpr int PER_PASS = BUCKET_COUNTS_THREADS / POISSON_G;
    for (int c0 = (int)blockIdx.x * PER_PASS; c0 < chunks; c0 += (int)gridDim.x * PER_PASS) {   // (uniform per workgroup)
        const int c = c0 + (int)threadIdx.x / POISSON_G;
        const bool active = c < chunks;
        double mu = 0.0;
        if (active) {
            const double mass = edges[lds_skew(c)] - (c > 0 ? edges[lds_skew(c - 1)] : 0.0);
            mu = (mass > 0.0 && total > 0.0) ? lambda * mass / total : 0.0;
        }
        const unsigned int x = poisson_draw(active, mu, (uint32_t)c, (epoch << 16), k0, k1, POISSON_G, lane & ~(POISSON_G - 1));
        if (active && (lane & (POISSON_G - 1)) == 0) {
            put_shared(&counts[c], x);
            put_shared(&extra[c], 0u);
        }
    }
    grid_barrier(bar, bar_base + gridDim.x);
    // ---- 2: the total, and this workgroup's share of the top-up ----
    unsigned long long mine = 0ull;
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS)
        mine += get_shared(&counts[c]);
    for (int off = QSMC_WAVE / 2; off > 0; off >>= 1) mine += __shfl_down(mine, off, QSMC_WAVE);
    if (lane == 0 && mine) atomicAdd(&total_s, mine);
    __syncthreads();
    const long long T = (long long)total_s;
    if (T < n_out) {
        const int64_t deficit = n_out - T, n_pairs = (deficit + 1) >> 1;
        for (int64_t pr = (int64_t)blockIdx.x * BUCKET_COUNTS_THREADS + threadIdx.x; pr < n_pairs;
             pr += (int64_t)gridDim.x * BUCKET_COUNTS_THREADS) {
            PhiloxStream rng{(uint64_t)pr, (epoch << 16), k0, k1};
            double u[2];
            rng.uniforms(3, u[0], u[1]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                if (2 * pr + e < deficit) {
                    int c = upper_bound_skew(edges, chunks, u[e]);     // #edges <= u == chunk index
                    if (c > chunks - 1) c = chunks - 1;                // u beyond cdf[n-1] (rounding): Q2 clamp
                    atomicAdd(&hist[c], 1u);
                }
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS)
            if (hist[c]) atomicAdd(&extra[c], hist[c]);
    }
    grid_barrier(bar, bar_base + 2ull * gridDim.x);
    if (blockIdx.x != 0) return;
    // ---- 3: final counts and the plan ----
    unsigned int *cnt = hist;
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS)
        cnt[c] = get_shared(&counts[c]) + get_shared(&extra[c]);
    __syncthreads();
    if (T > n_out) {                                             // (uniform branch; ~3e-7 of the resamples)
        if (threadIdx.x == 0) {
            long long left = T;
            for (long long i = 0; i < T - n_out; ++i, --left) {
                PhiloxStream rng{(uint64_t)i, (epoch << 16), k0, k1};
                double u, unused;
                rng.uniforms(4, u, unused);
                long long target = (long long)(u * (double)left);   // which of the remaining items goes
                if (target > left - 1) target = left - 1;
                int c = 0;
                for (long long run = (long long)cnt[0]; run <= target; run += (long long)cnt[c]) ++c;
                cnt[c] -= 1u;
            }
        }
        __syncthreads();
    }
    for (int c = threadIdx.x; c < chunks; c += BUCKET_COUNTS_THREADS) counts[c] = cnt[c];
    bucket_plan_block(cnt, chunks, cap, slot_off, item_off, item_chunk);
}

// counts[c] = sum_g hist[g][c].  A workgroup takes 64 chunks; its four waves each sum a quarter of the rows
// (coalesced 256-byte row segments), LDS combines them: 4x the workgroups and a quarter of the dependent
// loads per thread of the one-thread-per-chunk version (8.8 -> ~4 us, the launch floor).
__global__ __launch_bounds__(QSMC_BLOCK) void k_bucket_reduce(const unsigned int *__restrict__ hist, int rows,
                                                              int chunks, unsigned int *__restrict__ counts) {
    __shared__ unsigned int part[QSMC_WAVES_PER_BLOCK][QSMC_WAVE];
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int wave = threadIdx.x / QSMC_WAVE;
    const int c = blockIdx.x * QSMC_WAVE + lane;
    unsigned int s = 0;
    if (c < chunks) {
#pragma unroll 16
        for (int g = wave; g < rows; g += QSMC_WAVES_PER_BLOCK) s += hist[(size_t)g * chunks + c];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < chunks) {
#pragma unroll
        for (int wv = 1; wv < QSMC_WAVES_PER_BLOCK; ++wv) s += part[wv][lane];
        counts[c] = s;
    }
}

__global__ __launch_bounds__(1024) void k_bucket_plan(const unsigned int *__restrict__ counts, int chunks, int cap,
                                                      long long *__restrict__ slot_off,
                                                      int *__restrict__ item_off,
                                                      int *__restrict__ item_chunk) {
    bucket_plan_block(counts, chunks, cap, slot_off, item_off, item_chunk);
}

constexpr int BUCKET_RLIST_CAP = 1024;               // per-workgroup list of outputs that need a global redraw

// Draw + kick of one redraw round (round >= 1) of output slot o from the GLOBAL CDF.
template <int DM>
__device__ __forceinline__ bool redraw_rounds(int kind, int d, double min_freq, const double *__restrict__ x_in,
                                              int64_t ldx_in, int64_t n_in, const double *__restrict__ cdf,
                                              const LWArgs &lw, uint32_t k0, uint32_t k1, uint32_t epoch,
                                              int maxiter, int64_t o, double *p, const double *edges, int chunks) {
    for (int round = 1; round < maxiter; ++round) {
        PhiloxStream rng{(uint64_t)o, (epoch << 16) | (uint32_t)round, k0, k1};
        double u0, unused;
        rng.uniforms(0, u0, unused);
        int64_t j;
        if (edges) {
            // two levels: the chunk among the edges in LDS (the chunk's last CDF entry IS its upper edge, so
            // #edges <= u is the chunk of the upper bound), then 12 probes of that chunk's 32 KB of the CDF instead of
            // 24 scattered over all of it -- the same index as search_right over the whole table
            int c = upper_bound_skew(edges, chunks, u0);
            if (c > chunks - 1) c = chunks - 1;
            const int64_t base = (int64_t)c * SCAN_CHUNK;
            const int64_t len = n_in - base < SCAN_CHUNK ? n_in - base : SCAN_CHUNK;
            // (round 3 tried an interpolated starting point + galloping bracket here -- 3-4 neighbouring cache lines instead
            //  of 12 scattered ones, same index: no change, 145 vs 140 us per launch at a million redraws -- and a CDF-free
            //  rejection draw against each chunk's largest weight, which weights that are products of a dozen likelihoods
            //  make five times slower; tools/experiments/r3_redraw_by_rejection.patch)
            int64_t lo = 0, hi = len;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (cdf[base + mid] <= u0) lo = mid + 1; else hi = mid;
            }
            j = base + lo < n_in - 1 ? base + lo : n_in - 1;
        } else {
            j = search_right(cdf, n_in, u0);
        }
        double zz[DM];
#pragma unroll
        for (int q = 0; q < DM; q += 2) {
            if (q < d) {
                double z0, z1;
                rng.normals(1 + (q >> 1), z0, z1);
                zz[q] = z0;
                if (q + 1 < DM) zz[q + 1] = z1;
            }
        }
#pragma unroll
        for (int m = 0; m < DM; ++m) {
            if (m < d) {
                double s = 0.0;
#pragma unroll
                for (int q = 0; q < DM; ++q)
                    if (q < d) s += lw.S[m * d + q] * zz[q];
                p[m] = (lw.a * x_in[m * ldx_in + j] + (1.0 - lw.a) * lw.mean[m]) + s;
            }
        }
        if (model_valid(kind, p, min_freq)) return true;
    }
    return false;
}

// chunk_scan_block sink of the sampler: the entry goes to the skewed LDS table and, in the same pass,
// the guide table is filled -- entry j is the first one whose cell is >= k for every cell k in
// (cell(prev), cell(v)], so G[k] = j there (G[0] = 0, G[SGUIDE_BINS] = len): no histogram, no atomics,
// no extra barrier.  Runs longer than 8 cells (a dominant weight) are filled by the whole wave.
struct StoreLdsGuide {
    double *lcdf;
    unsigned short *G;
    double lo_edge, gscale;
    bool use_guide;                                // workgroup-uniform
    int len;
    int cp;                                        // cell of the previous entry (carried along the lane's run)
    __device__ __forceinline__ void operator()(int j, double v, double prev, bool live) {
        if (live) lcdf[lds_skew(j)] = v;
        if (!use_guide) return;
        const int lane = threadIdx.x & (QSMC_WAVE - 1);
        if ((j & (SCAN_PER_LANE - 1)) == 0) cp = j == 0 ? -1 : guide_cell<SGUIDE_BINS>(prev, lo_edge, gscale);
        const int cj = live ? guide_cell<SGUIDE_BINS>(v, lo_edge, gscale) : cp;
        const unsigned short js = (unsigned short)j;
        // straight-line for the common run lengths 0..4
        if (cj > cp) G[cp + 1] = js;
        if (cj > cp + 1) G[cp + 2] = js;
        if (cj > cp + 2) G[cp + 3] = js;
        if (cj > cp + 3) G[cp + 4] = js;
        if (live && j == len - 1) G[SGUIDE_BINS] = (unsigned short)len;
        unsigned long long long_runs = __ballot(cj > cp + 4);
        while (long_runs) {                        // a dominant weight: the wave fills the run together
            const int src = __ffsll((long long)long_runs) - 1;
            long_runs &= long_runs - 1;
            const int s0 = __shfl(cp + 5, src, QSMC_WAVE), e0 = __shfl(cj, src, QSMC_WAVE);
            const int jj = __shfl(j, src, QSMC_WAVE);
            for (int k = s0 + lane; k <= e0; k += QSMC_WAVE) G[k] = (unsigned short)jj;
        }
        cp = cj;
    }
};

// ---------------------------------------------------------------------------------------------
// The chunk's CDF as the samplers hold it in LDS (round 2).  Round 1 kept the 4096 entries in one skewed table with a
// 2048-cell guide over ALL entries, filled while the entries were stored: ~40 instructions per entry, two thirds of the
// set-up's 614 instructions per lane (24 us of the d = 1 kernel's 89 at N = 1e7).  The scan already groups the entries by
// lane -- lane l owns entries 8 l .. 8 l + 7 -- so now:
//   cdf8[4096]   the entries, unskewed: a lane's eight are 64 contiguous bytes;
//   tops[512]    entry 8 l + 7 of every lane (skewed index), i.e. the CDF at block granularity;
//   G[1025]      guide over the TOPS only (1024 cells over the chunk's mass: one guide entry per lane, not eight).
// A query finds its block among the tops (guide bracket, usually zero or one probe) and then counts the entries of
// that block that are <= u (four 16-byte LDS reads issued together, eight compares): the same index as an upper bound
// over all 4096 entries -- tops[s - 1] <= u < tops[s] puts every earlier block below u and every later one above.
// Entries past the end of a short last chunk read as +inf.
// ---------------------------------------------------------------------------------------------
// Round 5, late: a block of eight entries starts every 64 bytes, i.e. on LDS banks 0, 16, 32 or 48 -- the 64 random block
// reads of a wave's search land on four bank groups (SQ_LDS_BANK_CONFLICT 1.6e7 cycles per launch of the d = 1 sampler,
// as many as the LDS spends on its instructions), and the scan's stores (lane l -> block l) collide 16 ways.  With
// CDF8_PAD = 2 doubles between blocks a block starts every 80 bytes: banks 0, 20, 40, 60, 16, ... -- all sixteen
// 4-bank groups.  40 KB instead of 32; the kernels that can afford it take it (template parameter of the sink and of
// the search), k_bucket_anc16 keeps its three workgroups per CU instead.
template <int PAD> __device__ __forceinline__ int cdf8_pos(int j) { return j + PAD * (j >> 3); }
template <int PAD> constexpr int cdf8_size() { return BUCKET_CHUNK + PAD * (BUCKET_CHUNK / 8); }
constexpr int TGUIDE_BINS = 1024;
constexpr int CDF8_PAD = 2;
constexpr int TOPS_N = BUCKET_CHUNK / SCAN_PER_LANE;                 // 512
constexpr int TOPS_LDS = TOPS_N + (TOPS_N >> 5) + 4;
__device__ __forceinline__ int tops_skew(int l) { return l + (l >> 5); }

template <int PAD = 0>
struct StoreLdsTops {
    double *cdf8, *tops;
    unsigned short *G;
    double lo_edge, hi_edge, gscale;
    bool use_guide;                                // workgroup-uniform
    int len;
    int cp;                                        // guide cell of the previous block's top
    __device__ __forceinline__ void operator()(int j, double v, double prev, bool live) {
        cdf8[cdf8_pos<PAD>(j)] = live ? v : INFINITY;
        const int k = j & (SCAN_PER_LANE - 1), l = j >> 3;
        if (k == 0) cp = (j == 0 || !use_guide) ? -1 : guide_cell<TGUIDE_BINS>(prev, lo_edge, gscale);
        if (k != SCAN_PER_LANE - 1) return;
        const bool has = (j - (SCAN_PER_LANE - 1)) < len;          // the block holds at least one entry of the chunk
        const double top = live ? v : (has ? hi_edge : INFINITY);  // (a short block ends with the chunk's last entry = hi_edge)
        tops[tops_skew(l)] = top;
        if (!use_guide) return;
        const int lane = threadIdx.x & (QSMC_WAVE - 1);
        const int cj = has ? guide_cell<TGUIDE_BINS>(top, lo_edge, gscale) : cp;
        const unsigned short ls = (unsigned short)l;
        // block l is the first whose top's cell is >= c for every cell c in (cp, cj]: G[c] = l there
        if (cj > cp) G[cp + 1] = ls;
        if (cj > cp + 1) G[cp + 2] = ls;
        if (cj > cp + 2) G[cp + 3] = ls;
        if (cj > cp + 3) G[cp + 4] = ls;
        if (has && j + 1 >= len) G[TGUIDE_BINS] = (unsigned short)(l + 1);       // the chunk's last block: #blocks
        unsigned long long long_runs = __ballot(cj > cp + 4);
        while (long_runs) {                        // a dominant weight: the wave fills the run together
            const int src = __ffsll((long long)long_runs) - 1;
            long_runs &= long_runs - 1;
            const int s0 = __shfl(cp + 5, src, QSMC_WAVE), e0 = __shfl(cj, src, QSMC_WAVE);
            const int ll = __shfl(l, src, QSMC_WAVE);
            for (int c = s0 + lane; c <= e0; c += QSMC_WAVE) G[c] = (unsigned short)ll;
        }
    }
};

// number of entries of the chunk's CDF that are <= u (the caller clamps to len - 1)
template <int PAD = 0>
__device__ __forceinline__ int table_upper_bound(const double *cdf8, const double *tops, const unsigned short *G,
                                                 int n_blocks, bool use_guide, double lo_edge, double gscale, double u) {
    int lo = 0, hi = n_blocks;
    if (use_guide) {
        const int c = guide_cell<TGUIDE_BINS>(u, lo_edge, gscale);
        lo = G[c];
        hi = G[c + 1];
    }
    while (lo < hi) {                              // # tops <= u
        const int mid = (lo + hi) >> 1;
        if (tops[tops_skew(mid)] <= u) lo = mid + 1; else hi = mid;
    }
    const int sb = lo < n_blocks ? lo : n_blocks - 1;
    if (PAD == 0) {
        const double4 a = *reinterpret_cast<const double4 *>(cdf8 + 8 * sb);
        const double4 b = *reinterpret_cast<const double4 *>(cdf8 + 8 * sb + 4);
        const int cnt = (a.x <= u) + (a.y <= u) + (a.z <= u) + (a.w <= u) + (b.x <= u) + (b.y <= u) + (b.z <= u) + (b.w <= u);
        return 8 * sb + cnt;
    }
    const double2 *blk = reinterpret_cast<const double2 *>(cdf8 + (8 + PAD) * sb);     // (16-byte aligned: PAD is even)
    const double2 a = blk[0], b = blk[1], c = blk[2], d = blk[3];
    const int cnt = (a.x <= u) + (a.y <= u) + (b.x <= u) + (b.y <= u) + (c.x <= u) + (c.y <= u) + (d.x <= u) + (d.y <= u);
    return 8 * sb + cnt;
}

// The single-pass sampler (d <= 2).  One workgroup per work item.  The chunk's CDF is SCANNED HERE from the weights
// (bit-identical to k_chunk_scan), so the CDF never touches HBM; every output pair draws its positions, searches, gathers
// and is kicked in one loop iteration; a particle that fails postselection on its first try is queued for
// k_bucket_redraw, which alone needs the (then materialised) global CDF.  With one or two coordinates per particle the
// gather is cheap and the search's LDS round trips hide under the kick's arithmetic of the same iteration; from d = 3 on
// the ordered kernel below wins (measured at N = 1e7, d = 1: 89 us here, 104 us ordered; d = 3, N = 1.25e7: 315 vs 259).
// Occupancy: 49 KB of LDS (41 before the table was padded) allows three workgroups per CU; the small-d instantiations are held to 80
// VGPRs (6 waves/SIMD) so that the third one fits -- the kernel is VALU-issue bound and the extra
// waves hide the LDS search and gather latency (121 -> 110 us at N = 1e7, d = 1).
template <int D, int BT>   // D = 0: runtime d; BT = threads per workgroup
__attribute__((amdgpu_waves_per_eu(D >= 1 && D <= 2 ? 6 : 1, 8)))
__global__ __launch_bounds__(BT) void k_bucket_sample(
    int kind, int d_rt, double min_freq, int postselect, const double *__restrict__ x_in, int64_t ldx_in,
    int64_t n_in, const double *__restrict__ w, double inv_norm, const double *__restrict__ offsets,
    int chunks, const long long *__restrict__ slot_off,
    const int *__restrict__ item_off, const int *__restrict__ item_chunk, LWArgs lw, uint32_t k0, uint32_t k1,
    uint32_t epoch, int maxiter, double *__restrict__ x_out, OutPlace pl,
    unsigned long long *__restrict__ n_failed, unsigned int *__restrict__ retry_list,
    unsigned long long *__restrict__ retry_count, int cap) {
    constexpr int DM = D > 0 ? D : QSMC_MAX_D;
    const int d = D > 0 ? D : d_rt;
    constexpr int PAD = CDF8_PAD;
    __shared__ __attribute__((aligned(32))) double lcdf[cdf8_size<PAD>()];
    __shared__ double ltops[TOPS_LDS];
    __shared__ unsigned short lguide[TGUIDE_BINS + 2];
    __shared__ double wave_tot[SCAN_WAVES];
    __shared__ unsigned short rlist[BUCKET_RLIST_CAP];          // slot - o_begin < BUCKET_CAP
    static_assert(BT >= SCAN_THREADS, "the in-sampler chunk scan needs 512 threads");
    __shared__ int rcount;
    __shared__ unsigned long long rbase;
    if ((int)blockIdx.x >= item_off[chunks]) return;
    const int c = item_chunk[blockIdx.x];
    const int part = (int)blockIdx.x - item_off[c];
    const long long slot0 = slot_off[c], n_c = slot_off[c + 1] - slot0;
    const long long t0 = (long long)part * cap;
    const long long t1 = t0 + cap < n_c ? t0 + cap : n_c;
    const int64_t base = (int64_t)c * BUCKET_CHUNK;
    const int len = (int)((n_in - base) < BUCKET_CHUNK ? (n_in - base) : BUCKET_CHUNK);
    if (threadIdx.x == 0) rcount = 0;
    const double lo_edge = chunk_edge(offsets, c);
    const double hi_edge = offsets[c + 1];
    const double gscale = (double)TGUIDE_BINS / (hi_edge - lo_edge);
    const bool use_guide = hi_edge > lo_edge && gscale < 1e300;     // workgroup-uniform
    chunk_scan_block(w, n_in, inv_norm, offsets, (int64_t)c, wave_tot,
                     StoreLdsTops<PAD>{lcdf, ltops, lguide, lo_edge, hi_edge, gscale, use_guide, len, -1});
    __syncthreads();
    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;         // this item's output slots
    unsigned long long failed = 0;
    // pairs of output slots (2P, 2P+1) share their Philox blocks; a pair straddling two work items is
    // evaluated by both, each writing only its own half.
    // Stage A (ancestors): position Philox -> guided LDS search -> gather of x (d <= 4: into registers).
    // Stage B (kick): normals Philox + Box-Muller -> Liu-West combine -> validity -> store.
    // A is issued first so that its LDS round trips and the L2/HBM gather are in flight during B's ~600
    // cycles of independent arithmetic (110 -> 100 us at N = 1e7; issuing A of the NEXT pair ahead of B --
    // a software pipeline -- was measured too and is slower, 115 us: spills).  Both halves of a pair are
    // searched even if one belongs to the neighbouring work item: no divergence, cheap.
    // d = 16 (2-qubit tomography, BIG): (i) the 32 ancestor coordinates of a pair are gathered in stage A as well,
    // in flight during the kick's arithmetic instead of between its products; (ii) S z runs as 8 + 8 steps of a
    // RUN-TIME loop -- draw one Box-Muller pair, add its two columns of S into the 16 sums -- so that only two
    // columns of S (scalar loads at a run-time offset) and two normals are live: fully unrolled, all 256 entries
    // of S sat in SGPRs spilled to VGPR lanes (2200 v_readlane / v_writelane per pair).  Same products, same order
    // of additions: bit-identical particles.  413 -> 386 us at N = 1.25e6 -- modest, because the kernel is simply
    // heavy: 16 Box-Muller normals and a 16 x 16 product per output, ~12k issue cycles at two waves per SIMD.
    constexpr bool BIG = (D == 16);
    constexpr bool EARLY = DM <= 4 || BIG;
    struct Anc {
        int jl[2];
        double xg[2][EARLY ? DM : 1];
    };
    auto stage_a = [&](int64_t P, Anc &an) {
        PhiloxStream rng{(uint64_t)P, (epoch << 16), k0, k1};
        double upos[2];
        rng.uniforms(1, upos[0], upos[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // position inside this chunk: given the counts, uniform on [lo_edge, hi_edge)
            const double u = lo_edge + upos[e] * (hi_edge - lo_edge);
            int j = table_upper_bound<PAD>(lcdf, ltops, lguide, (len + 7) >> 3, use_guide, lo_edge, gscale, u);
            an.jl[e] = j > len - 1 ? len - 1 : j;
            if (EARLY) {
#pragma unroll
                for (int m = 0; m < DM; ++m)
                    if (m < d) an.xg[e][m] = x_in[m * ldx_in + base + an.jl[e]];
            }
        }
    };
    auto stage_b = [&](int64_t P, const Anc &an) {
        double z[BIG ? 2 : 2 * DM];
        double acc[BIG ? 2 : 1][BIG ? DM : 1];
        PhiloxStream nrm{0, (epoch << 16), k0, k1};
        if (BIG) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
#pragma unroll
                for (int m = 0; m < DM; ++m) acc[e][BIG ? m : 0] = 0.0;
#pragma unroll 1
                for (int kq = 0; kq < DM / 2; ++kq) {            // normals 2 kq, 2 kq + 1 of output e
                    nrm.particle = (uint64_t)P * (uint64_t)DM + (uint64_t)(e * (DM / 2) + kq);
                    nrm.normals(2, z[0], z[1]);
                    const int q = 2 * kq;
#pragma unroll
                    for (int m = 0; m < DM; ++m) {
                        double t = acc[e][BIG ? m : 0];
                        t += lw.S[m * DM + q] * z[0];
                        t += lw.S[m * DM + q + 1] * z[1];
                        acc[e][BIG ? m : 0] = t;
                    }
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < DM; ++k) {
                if (k < d) {
                    nrm.particle = (uint64_t)P * (uint64_t)d + (uint64_t)k;
                    nrm.normals(2, z[BIG ? 0 : 2 * k], z[BIG ? 1 : 2 * k + 1]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t o = 2 * P + e;
            if (o >= o_begin && o < o_end) {
                double p[DM];
#pragma unroll
                for (int m = 0; m < DM; ++m) {
                    if (m < d) {
                        double sm = 0.0;
                        if (BIG) {
                            sm = acc[BIG ? e : 0][BIG ? m : 0];
                        } else {
#pragma unroll
                            for (int q = 0; q < DM; ++q)
                                if (q < d) sm += lw.S[m * d + q] * z[BIG ? 0 : e * d + q];
                        }
                        const double xa = EARLY ? an.xg[e][m] : x_in[m * ldx_in + base + an.jl[e]];
                        p[m] = (lw.a * xa + (1.0 - lw.a) * lw.mean[m]) + sm;
                    }
                }
                bool ok = !postselect || model_valid(kind, p, min_freq);
                if (!ok && maxiter > 1) {
                    // queue for k_bucket_retry (needs the global CDF)
                    const int idx = atomicAdd(&rcount, 1);
                    if (idx < BUCKET_RLIST_CAP) rlist[idx] = (unsigned short)(o - o_begin);
                    else retry_list[atomicAdd(retry_count, 1ull)] = (unsigned int)o;   // rare overflow path
                    ok = true;                      // decided later
                }
                const int64_t row = place_row(pl, o);
#pragma unroll
                for (int m = 0; m < DM; ++m)
                    if (m < d) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
                if (!ok) ++failed;
            }
        }
    };
    for (int64_t P = (o_begin >> 1) + threadIdx.x; 2 * P < o_end; P += BT) {
        Anc an;
        stage_a(P, an);
        stage_b(P, an);
    }
    if (failed) atomicAdd(n_failed, failed);
    __syncthreads();
    const int nl = rcount < BUCKET_RLIST_CAP ? rcount : BUCKET_RLIST_CAP;
    if (nl == 0) return;
    if (threadIdx.x == 0) rbase = atomicAdd(retry_count, (unsigned long long)nl);   // one atomic per workgroup
    __syncthreads();
    for (int i = threadIdx.x; i < nl; i += BT) retry_list[rbase + i] = (unsigned int)(o_begin + rlist[i]);
}

// ---------------------------------------------------------------------------------------------
// The ordered sampler (d >= 3).  One workgroup per work item = (chunk c, <= cap of its output slots), three phases:
//   1  ANCESTORS.  The chunk's CDF is scanned here from the weights into LDS (bit-identical to k_chunk_scan, so the CDF
//      never touches HBM) together with its guide table; every output slot draws its position inside the chunk (given
//      the counts, positions are i.i.d. uniform on the chunk's mass: exact), finds its ancestor by the guided LDS
//      search and adds one to that particle's counter (two 16-bit counters per LDS word).
//   2  ORDER.  An integer scan of the 4096 counters; each particle writes its index as many times as it was drawn:
//      the item's ancestors in ASCENDING order (the list overlays the CDF, which is no longer needed).  Outputs are
//      exchangeable, so slot o_begin + k may take the k-th smallest ancestor -- and then neighbouring lanes gather
//      neighbouring particles: the gather of x was 64 distinct cache lines per wave instruction per coordinate when
//      the ancestors came in drawing order, now it is a handful.
//   3  KICK.  Slots in pairs (2 P, 2 P + 1) sharing the Box-Muller pairs of Philox blocks P d + m (slot 2), Liu-West
//      combine, validity, store; a particle that fails postselection is queued for k_bucket_redraw, which alone needs
//      the (then materialised) global CDF.  The loop has no search and no dependent LDS chain in it any more.
// A pair straddling two work items is evaluated by both, each writing only its own half.
// Occupancy: 59 KB of LDS (51 before the table was padded), 92 - 106 VGPRs at d = 3, 4: two workgroups per CU.
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// Proposal bank (round 3): postselection without global redraws, for models whose constraint bites at EVERY resample
// (RB: A + B <= 1 runs through the cloud; 8 % of the first-try children fail at N = 1.25e7, a million of them).
// Round 2 sent each of them back to the global CDF: a 100 MB scan + write (k_chunk_scan, 75 us) and a redraw kernel of
// scattered searches and gathers (k_bucket_redraw, 145-170 us, 717 MB of line traffic) per resample.
// A redraw is nothing but a FRESH proposal -- ancestor ~ w, kick -- so the sampler now also produces a bank of spare
// proposals while every chunk's CDF is in LDS anyway, and a failed slot takes bank entries until one is valid:
//   * how many spares each work item makes is Poisson(lambda mass_item / total), independently per item: given their
//     total E the spares' ancestors are then i.i.d. ~ w (Poissonisation again), whatever E turns out to be;
//   * a spare = position in the chunk (Philox round tag 0xFFFF, block (item, pair), slot 1) -> LDS search -> gather ->
//     kick (normals: round tag 0xFFFF, blocks ((item << 12 | pair) d + q, slot 2)) -> validity flag, written to the
//     item's reservation in the bank (one atomic per item; WHERE the reservation lands is timing, WHAT it holds is not);
//   * the failed first tries of an item are listed in ascending slot order (a bit mask in LDS, not an atomic append),
//     so "the j-th failed slot" and "the g-th spare" (items in order, spares in pair order) mean the same on every run;
//   * k_bank_round t = 1, 2, ...: the j-th slot still failed takes spare perm(B_t + j), perm a keyed bijection of
//     [0, E) (a fixed one would hand the early chunks' spares to the early chunks' failures; which spares are consumed
//     must not depend on where they came from), B_t = spares handed out before round t.  Valid: the slot is done.
//     Invalid: it is listed again (block-local compaction + k_bank_scan's prefix of the block counts) for round t + 1.
//     Each spare is looked at once, each failed slot gets independent proposals: the law of the redraw loop
//     (resamplers.py:341-372), without its search.  BANK_ROUNDS rounds are queued (8 % -> 0.6 % -> ... a million
//     failures are through in six); slots left after that, or past the end of the bank, go to k_bucket_redraw.
// Deterministic for given Philox keys; oracle/philox.py (bank_*) mirrors it entry for entry.
// ---------------------------------------------------------------------------------------------
// doubles per bank entry: x[0..d) and, last, the validity flag: 4 for d <= 3, 8 for d = 4 (entries stay 32 / 64-byte aligned)
__host__ __device__ __forceinline__ int bank_stride(int d) { return d <= 3 ? 4 : 8; }
constexpr int BANK_MAX_PER_ITEM = 4096;              // spares per work item (8 per thread, kept in registers between phases)
constexpr int BANK_ROUNDS = 7;
constexpr int BANK_VB = 512;                         // entries per (virtual) block of a round
struct BankOut {
    double lambda;                                   // expected spares over the whole launch; 0: no bank
    int stride, reserved;                            // bank_stride(d)
    double *entries;                                 // [capacity][BANK_STRIDE]
    long long capacity;
    int *e_cnt, *f_cnt;                              // per work item: spares to make (k_bank_counts) / first tries that failed
    long long *e_base, *f_base;                      // per work item: where its spares (prefix of e_cnt) / its failed slots start
};

constexpr int SMP_HEAVY = 24, SMP_HEAVY_CAP = BUCKET_CAP / SMP_HEAVY + 8;

template <int D, int BT>   // D = 0: runtime d; BT = threads per workgroup.  (Held to 80 VGPRs for a third workgroup per
                           // CU, d = 3 spills 14 registers and is slower: 277 vs 259 us.)
__global__ __launch_bounds__(BT) void k_bucket_sample_ordered(
    int kind, int d_rt, double min_freq, int postselect, const double *__restrict__ x_in, int64_t ldx_in,
    int64_t n_in, const double *__restrict__ w, double inv_norm, const double *__restrict__ offsets,
    int chunks, const long long *__restrict__ slot_off,
    const int *__restrict__ item_off, const int *__restrict__ item_chunk, LWArgs lw, uint32_t k0, uint32_t k1,
    uint32_t epoch, int maxiter, double *__restrict__ x_out, OutPlace pl,
    unsigned long long *__restrict__ n_failed, unsigned int *__restrict__ retry_list,
    unsigned long long *__restrict__ retry_count, int cap, BankOut bank) {
    constexpr int DM = D > 0 ? D : QSMC_MAX_D;
    const int d = D > 0 ? D : d_rt;
    constexpr int PAD = CDF8_PAD;
    __shared__ __attribute__((aligned(32))) double lcdf[cdf8_size<PAD>()];
    __shared__ double ltops[TOPS_LDS];
    __shared__ unsigned short lguide[TGUIDE_BINS + 2];
    __shared__ unsigned int cnt2[BUCKET_CHUNK / 2];             // children per source particle, two counters a word
    __shared__ double wave_tot[SCAN_WAVES];
    __shared__ int iwave_tot[SCAN_WAVES];
    __shared__ unsigned int heavy[2 * SMP_HEAVY_CAP];
    __shared__ unsigned int failmask[BUCKET_CAP / 32];          // bit (slot - o_begin): the first try failed postselection
    static_assert(BT == SCAN_THREADS, "one lane owns 8 consecutive source particles: 512 threads per chunk");
    static_assert(BUCKET_CAP * 2 <= BUCKET_CHUNK * 8, "the ancestor list overlays the CDF");
    static_assert(BUCKET_CAP / 32 <= BT, "one thread per word of the failure mask");
    __shared__ int hcount;
    __shared__ long long rbase, ebase_s;
    const bool banked = bank.lambda > 0.0;                      // (uniform)
    if ((int)blockIdx.x >= item_off[chunks]) {
        if (banked && threadIdx.x == 0) bank.f_cnt[blockIdx.x] = 0;
        return;
    }
    const int c = item_chunk[blockIdx.x];
    const int part = (int)blockIdx.x - item_off[c];
    const long long slot0 = slot_off[c], n_c = slot_off[c + 1] - slot0;
    const long long t0 = (long long)part * cap;
    const long long t1 = t0 + cap < n_c ? t0 + cap : n_c;
    const int64_t base = (int64_t)c * BUCKET_CHUNK;
    const int len = (int)((n_in - base) < BUCKET_CHUNK ? (n_in - base) : BUCKET_CHUNK);
    if (threadIdx.x == 0) hcount = 0;
    if (threadIdx.x < BUCKET_CAP / 32) failmask[threadIdx.x] = 0u;
    for (int k = threadIdx.x; k < BUCKET_CHUNK / 2; k += BT) cnt2[k] = 0u;
    const double lo_edge = chunk_edge(offsets, c);
    const double hi_edge = offsets[c + 1];
    const double gscale = (double)TGUIDE_BINS / (hi_edge - lo_edge);
    const bool use_guide = hi_edge > lo_edge && gscale < 1e300;     // workgroup-uniform
    chunk_scan_block(w, n_in, inv_norm, offsets, (int64_t)c, wave_tot,
                     StoreLdsTops<PAD>{lcdf, ltops, lguide, lo_edge, hi_edge, gscale, use_guide, len, -1});
    __syncthreads();
    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;         // this item's output slots
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    // ---- 1: ancestors, histogrammed.  Position of slot o: word o & 1 of Philox block (o >> 1, round 0, slot 1)
    for (int64_t P = (o_begin >> 1) + threadIdx.x; 2 * P < o_end; P += BT) {
        PhiloxStream rng{(uint64_t)P, (epoch << 16), k0, k1};
        double upos[2];
        rng.uniforms(1, upos[0], upos[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t o = 2 * P + e;
            const double u = lo_edge + upos[e] * (hi_edge - lo_edge);
            int j = table_upper_bound<PAD>(lcdf, ltops, lguide, (len + 7) >> 3, use_guide, lo_edge, gscale, u);
            j = j > len - 1 ? len - 1 : j;
            if (o >= o_begin && o < o_end) atomicAdd(&cnt2[j >> 1], 1u << (16 * (j & 1)));
        }
    }
    // ---- 1x: the spares of the proposal bank (e_i of them: k_bank_counts).  Their ancestors are found now, while the CDF
    // is still in LDS, and wait in registers (two 16-bit indices a word) until the primaries have been kicked.
    int e_i = 0;
    unsigned int jx[BANK_MAX_PER_ITEM / (2 * BT)];
    if (banked) {
        e_i = bank.e_cnt[blockIdx.x];                        // (k_bank_counts drew it: the Poisson sampler inlined here cost
                                                              //  27 VGPRs and a wave of occupancy)
        if (threadIdx.x == 0) {
            // where this item's spares go: the prefix of the counts (k_bank_counts) -- a reservation by a returning atomic
            // on one word, 4600 of them, was a visible part of the kernel's time
            long long eb = bank.e_base[blockIdx.x];
            if (eb + e_i > bank.capacity) eb = -1;             // (cannot happen with the host's sizing; the item then banks nothing)
            ebase_s = eb;
        }
#pragma unroll
        for (int k = 0; k < BANK_MAX_PER_ITEM / (2 * BT); ++k) {
            const int P = (int)threadIdx.x + k * BT;            // spare pair P: spares 2 P, 2 P + 1 of this item
            jx[k] = 0u;
            if (2 * P < e_i) {
                PhiloxStream rng{((uint64_t)blockIdx.x << 12) | (uint64_t)P, (epoch << 16) | 0xFFFFu, k0, k1};
                double upos[2];
                rng.uniforms(1, upos[0], upos[1]);
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const double u = lo_edge + upos[e] * (hi_edge - lo_edge);
                    int j = table_upper_bound<PAD>(lcdf, ltops, lguide, (len + 7) >> 3, use_guide, lo_edge, gscale, u);
                    j = j > len - 1 ? len - 1 : j;
                    jx[k] |= (unsigned int)j << (16 * e);
                }
            }
        }
    }
    __syncthreads();
    if (banked && ebase_s < 0) e_i = 0;                      // (E is clamped to the capacity by k_bank_counts: never addressed)
    // ---- 2: the ancestors in ascending order (lane l owns particles 8 l .. 8 l + 7; the list overlays the CDF)
    unsigned short *sorted = reinterpret_cast<unsigned short *>(lcdf);
    auto expand_sorted = [&]() {              // cnt2 (children per particle) -> sorted[0 .. total): workgroup-wide, ends on a barrier
        const int j0 = (int)threadIdx.x * 8;
        int nj[8], lt = 0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const unsigned int two = cnt2[(j0 + k) >> 1];
            nj[k] = (int)(two & 0xffffu);
            nj[k + 1] = (int)(two >> 16);
            lt += nj[k] + nj[k + 1];
        }
        int inc = lt;
#pragma unroll
        for (int off = 1; off < QSMC_WAVE; off <<= 1) {
            const int t = __shfl_up(inc, off, QSMC_WAVE);
            if (lane >= off) inc += t;
        }
        if (lane == QSMC_WAVE - 1) iwave_tot[wave] = inc;
        __syncthreads();                                            // (also: every CDF read of phase 1 is done)
        int off0 = inc - lt;
        for (int wv = 0; wv < wave; ++wv) off0 += iwave_tot[wv];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int nk = nj[k];
            if (nk > SMP_HEAVY) {                                    // a dominant particle: the whole workgroup fills its run
                const int h = atomicAdd(&hcount, 1);
                heavy[2 * h] = (unsigned int)off0 | ((unsigned int)(j0 + k) << 16);
                heavy[2 * h + 1] = (unsigned int)nk;
            } else {
                for (int r = 0; r < nk; ++r) sorted[off0 + r] = (unsigned short)(j0 + k);
            }
            off0 += nk;
        }
        __syncthreads();
        const int nh = hcount;                                      // (uniform: read behind the barrier, not written again)
        if (nh) {
            for (int h = 0; h < nh; ++h) {
                const unsigned int st = heavy[2 * h] & 0xffffu, jj = heavy[2 * h] >> 16, cn = heavy[2 * h + 1];
                for (unsigned int r = threadIdx.x; r < cn; r += BT) sorted[st + r] = (unsigned short)jj;
            }
            __syncthreads();
        }
    };
    expand_sorted();
    // ---- 3: the kicks
    unsigned long long failed = 0;
    constexpr bool EARLY = DM <= 4;
    for (int64_t P = (o_begin >> 1) + threadIdx.x; 2 * P < o_end; P += BT) {
        int jl[2];
        double xg[2][EARLY ? DM : 1];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t o = 2 * P + e;
            jl[e] = (o >= o_begin && o < o_end) ? (int)sorted[o - o_begin] : 0;
            if (EARLY) {                                            // (in flight during the normals' arithmetic)
#pragma unroll
                for (int m = 0; m < DM; ++m)
                    if (m < d) xg[e][m] = x_in[m * ldx_in + base + jl[e]];
            }
        }
        double z[2 * DM];
        PhiloxStream nrm{0, (epoch << 16), k0, k1};
#pragma unroll
        for (int k = 0; k < DM; ++k) {
            if (k < d) {
                nrm.particle = (uint64_t)P * (uint64_t)d + (uint64_t)k;
                nrm.normals(2, z[2 * k], z[2 * k + 1]);
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t o = 2 * P + e;
            if (o >= o_begin && o < o_end) {
                double p[DM];
#pragma unroll
                for (int m = 0; m < DM; ++m) {
                    if (m < d) {
                        double sm = 0.0;
#pragma unroll
                        for (int q = 0; q < DM; ++q)
                            if (q < d) sm += lw.S[m * d + q] * z[e * d + q];
                        const double xa = EARLY ? xg[e][m] : x_in[m * ldx_in + base + jl[e]];
                        p[m] = (lw.a * xa + (1.0 - lw.a) * lw.mean[m]) + sm;
                    }
                }
                bool ok = !postselect || model_valid(kind, p, min_freq);
                if (!ok && maxiter > 1) {
                    // listed for the proposal bank / k_bucket_redraw: a bit per slot, so that the list comes out in slot order
                    atomicOr(&failmask[(unsigned int)(o - o_begin) >> 5], 1u << ((unsigned int)(o - o_begin) & 31u));
                    ok = true;                      // decided later
                }
                const int64_t row = place_row(pl, o);
#pragma unroll
                for (int m = 0; m < DM; ++m)
                    if (m < d) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
                if (!ok) ++failed;
            }
        }
    }
    // ---- 3x: the spares are kicked like any slot (their own normals) and banked with their validity.  Spare k is kicked
    // from ITS OWN ancestor, in drawing order (round 5; until then the spares went through a second histogram ->
    // ascending list like the primaries: a counter clear, an LDS atomic per spare, an integer scan and six workgroup
    // barriers per work item, ~2 us each, to coalesce the gathers of the ~270 spares an item makes -- which hit lines
    // the primaries have just pulled into the L2 anyway).
    if (banked && e_i > 0) {
        const long long eb = ebase_s;
#pragma unroll
        for (int k = 0; k < BANK_MAX_PER_ITEM / (2 * BT); ++k) {
            const int P = (int)threadIdx.x + k * BT;
            if (2 * P < e_i) {
                double z[2 * DM];
                PhiloxStream nrm{0, (epoch << 16) | 0xFFFFu, k0, k1};
#pragma unroll
                for (int q = 0; q < DM; ++q) {
                    if (q < d) {
                        nrm.particle = (((uint64_t)blockIdx.x << 12) | (uint64_t)P) * (uint64_t)d + (uint64_t)q;
                        nrm.normals(2, z[2 * q], z[2 * q + 1]);
                    }
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (2 * P + e < e_i) {
                        const int jl = (int)((jx[k] >> (16 * e)) & 0xffffu);
                        double *ent = bank.entries + (eb + 2 * P + e) * bank.stride;
                        double p[DM];
#pragma unroll
                        for (int m = 0; m < DM; ++m) {
                            if (m < d) {
                                double sm = 0.0;
#pragma unroll
                                for (int q = 0; q < DM; ++q)
                                    if (q < d) sm += lw.S[m * d + q] * z[e * d + q];
                                p[m] = (lw.a * x_in[m * ldx_in + base + jl] + (1.0 - lw.a) * lw.mean[m]) + sm;
                                if (m < bank.stride - 1) ent[m] = p[m];
                            }
                        }
                        ent[bank.stride - 1] = model_valid(kind, p, min_freq) ? 1.0 : 0.0;
                    }
                }
            }
        }
    }
    if (failed) atomicAdd(n_failed, failed);
    __syncthreads();
    // the failed first tries, in ascending slot order: word t of the mask belongs to thread t
    {
        const unsigned int word = threadIdx.x < BUCKET_CAP / 32 ? failmask[threadIdx.x] : 0u;
        const int pc = __popc(word);
        int inc = pc;
#pragma unroll
        for (int off = 1; off < QSMC_WAVE; off <<= 1) {
            const int t = __shfl_up(inc, off, QSMC_WAVE);
            if (lane >= off) inc += t;
        }
        if (lane == QSMC_WAVE - 1) iwave_tot[wave] = inc;
        __syncthreads();
        int off0 = inc - pc, nl = 0;
        for (int wv = 0; wv < SCAN_WAVES; ++wv) {
            if (wv < wave) off0 += iwave_tot[wv];
            nl += iwave_tot[wv];
        }
        if (threadIdx.x == 0) {
            rbase = nl ? (long long)atomicAdd(retry_count, (unsigned long long)nl) : 0ll;      // one atomic per workgroup
            if (banked) {
                bank.f_cnt[blockIdx.x] = nl;
                bank.f_base[blockIdx.x] = rbase;
            }
        }
        __syncthreads();
        unsigned int bits = word;
        long long at = rbase + off0;
        while (bits) {
            const int bpos = __ffs((int)bits) - 1;
            bits &= bits - 1u;
            retry_list[at++] = (unsigned int)(o_begin + 32 * (int)threadIdx.x + bpos);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The bank's consumer: see the "Proposal bank" block above k_bucket_sample_ordered.
// ctr layout (long long): [0] E, [1] F (failed first tries), [2] leftover count (unsigned, bumped atomically),
//                         [8 + t] F_t, [24 + t] B_t for round t = 1 .., [40 + t] workgroups of round t that are done
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_I_STAGE = 8192;             // entries block_exclusive_scan_i takes in one sweep (through LDS), by default
template <bool ATOMIC, int STAGE = SCAN_I_STAGE>
__device__ __forceinline__ long long block_exclusive_scan_i(const int *src, int m, long long *__restrict__ dst);

// How many spares each work item makes: e_i ~ Poisson(lambda mass_c / total / parts_c), independently per item -- the
// parts of a split chunk share its mass evenly (independent Poissons add up to the chunk's).  Four lanes per item
// (poisson_draw's group), block (item, attempt) of round tag 0xFFFE.  Launched between the plan and the sampler.
__global__ __launch_bounds__(QSMC_BLOCK) void k_bank_counts(const double *__restrict__ offsets, int chunks,
                                                            const int *__restrict__ item_off, const int *__restrict__ item_chunk,
                                                            int max_items, double lambda, uint32_t k0, uint32_t k1, uint32_t epoch,
                                                            int *e_cnt, long long *__restrict__ e_off, long long *ctr,
                                                            long long capacity) {
    __shared__ int is_last;
    const int lane = threadIdx.x & (QSMC_WAVE - 1);
    const int n_items = item_off[chunks];
    const double total = offsets[chunks];
    for (int i0 = (int)blockIdx.x * (QSMC_BLOCK / POISSON_G); i0 < max_items; i0 += (int)gridDim.x * (QSMC_BLOCK / POISSON_G)) {
        const int it = i0 + (int)threadIdx.x / POISSON_G;
        const bool active = it < n_items;
        double mu = 0.0;
        if (active) {
            const int c = item_chunk[it];
            const int parts = item_off[c + 1] - item_off[c];
            const double lo_edge = chunk_edge(offsets, c), hi_edge = offsets[c + 1];
            mu = (total > 0.0 && hi_edge > lo_edge) ? lambda * ((hi_edge - lo_edge) / total) / (double)parts : 0.0;
        }
        unsigned int e = poisson_draw(active, mu, (uint32_t)it, (epoch << 16) | 0xFFFEu, k0, k1, POISSON_G, lane & ~(POISSON_G - 1));
        e = e < (unsigned)BANK_MAX_PER_ITEM ? e : (unsigned)BANK_MAX_PER_ITEM;
        if (it < max_items && (lane & (POISSON_G - 1)) == 0)
            __hip_atomic_store(&e_cnt[it], active ? (int)e : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the last workgroup to finish forms the prefix: every item's place in the bank is known before the sampler starts
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long done = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(ctr + 40), 1ull,
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = done == (unsigned long long)gridDim.x - 1ull;
    }
    __syncthreads();
    if (!is_last) return;
    const long long E = block_exclusive_scan_i<true>(e_cnt, n_items, e_off);
    if (threadIdx.x == 0) ctr[0] = E < capacity ? E : capacity;
}

struct BankIn {
    const double *entries;
    int stride, reserved;
    const int *e_cnt, *f_cnt;
    const long long *f_base;                          // per work item: where its failed slots start in retry_list
    long long *e_off, *f_off;                         // [max_items + 1] exclusive prefixes over the work items (an item's
                                                      //  spares sit AT e_off[item] in the bank: entry g of the bank is spare g)
    long long *ctr;                                   // [0] E, [1] F, [2] leftover count, [8 + t] F_t, [24 + t] B_t, [40 + t] blocks done
    unsigned int *tmp[2];                             // failed slots of a round, virtual block by virtual block
    int *bcount[2];                                   // ... how many in each virtual block
    long long *boff[2];                               // ... and their exclusive prefix
    int *vb_first[2];                                 // round t, virtual block vb: the entry of the prefix round t reads (f_off
                                                      //  for t = 1, boff[(t - 1) & 1] after) that holds position vb * BANK_VB
    unsigned int *leftover;                           // slots for k_bucket_redraw
    unsigned long long key[4];                        // of the bijection
};

// a keyed bijection of [0, n): four rounds of (odd multiply, add key, xor-shift) on ceil(log2 n) bits, cycle-walked
__host__ __device__ __forceinline__ unsigned long long bank_perm(unsigned long long g, unsigned long long n,
                                                                 const unsigned long long *key) {
    if (n < 2ull) return 0ull;
    int b = 1;
    while ((1ull << b) < n) ++b;
    const unsigned long long mask = (1ull << b) - 1ull;
    const int s1 = b / 2 > 0 ? b / 2 : 1, s2 = (b + 2) / 3 > 0 ? (b + 2) / 3 : 1;
    unsigned long long x = g;
    do {
        x = (x * 0x9E3779B97F4A7C15ull + key[0]) & mask;
        x ^= x >> s1;
        x = (x * 0xBF58476D1CE4E5B9ull + key[1]) & mask;
        x ^= x >> s2;
        x = (x * 0x94D049BB133111EBull + key[2]) & mask;
        x ^= x >> s1;
        x = (x * 0xD6E8FEB86659FD93ull + key[3]) & mask;
        x ^= x >> s2;
    } while (x >= n);
    return x;
}

// number of entries of the non-decreasing a[0..m) that are <= v
__device__ __forceinline__ int upper_bound_ll(const long long *__restrict__ a, int m, long long v) {
    int lo = 0, hi = m;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// Round 5: where a round's positions are, without a binary search through L2.  The j-th slot still failed is found through
// a prefix (per work item before round 1, per virtual block of the previous round after): 12-13 DEPENDENT loads per
// thread, and with the (redundant) search for the spare's work item behind it a try was a chain of ~30 -- k_bank_round
// was 33-72 us for a million tries that move 60 MB.  The workgroup that forms a prefix now also notes, for every block
// of BANK_VB consecutive positions, the entry its first position falls in (bank_block_starts: no search, every entry
// writes the blocks that start inside it); a round's workgroup loads the BT + 1 prefix entries from there on into LDS
// in one trip and searches those (bank_locate) -- the global search remains for a block whose positions run past the
// window (more than BT entries with next to no failures each).  Same entry, so the same slots in the same order.
__device__ __forceinline__ void bank_block_starts(const long long *off, int m, int *__restrict__ vb_first) {
    for (int i = threadIdx.x; i < m; i += (int)blockDim.x) {
        const long long lo = off[i], hi = off[i + 1];
        for (long long vb = (lo + BANK_VB - 1) / BANK_VB; vb * BANK_VB < hi; ++vb) vb_first[vb] = i;
    }
}

// entry i with off[i] <= j < off[i + 1] (off[0..m], off[m] = total > j); first = the entry of the block's first position.
// Workgroup-wide (one barrier inside; the caller has one between two calls); s_win holds BT + 1 entries.
template <int BT>
__device__ __forceinline__ int bank_locate(const long long *__restrict__ off, int m, int first, long long j, bool live,
                                           long long *s_win, long long &off_i) {
    for (int t = threadIdx.x; t <= BT; t += BT) s_win[t] = first + t <= m ? off[first + t] : 0x7fffffffffffffffll;
    __syncthreads();
    off_i = 0ll;
    if (!live) return 0;
    if (s_win[BT] <= j) {
        const int i = upper_bound_ll(off, m + 1, j) - 1;
        off_i = off[i];
        return i;
    }
    int lo = 0, hi = BT + 1;                          // # window entries <= j (s_win[0] <= j: at least one)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (s_win[mid] <= j) lo = mid + 1; else hi = mid;
    }
    off_i = s_win[lo - 1];
    return first + lo - 1;
}

// one workgroup: exclusive prefix of src[0..m) into dst[0..m], total returned to every thread.  ATOMIC: src was written
// by other workgroups of THIS launch with agent-scope atomic stores (the XCDs' L2s are not coherent inside a launch).
// STAGE: the static LDS the sweep takes (4 bytes an entry) -- every kernel that inlines this pays it whether or not it
// is the workgroup that scans: k_bank_round, a latency-bound gather where only the last workgroup scans, takes 2048 (8 KB;
// its lists are the failed slots' virtual blocks: <= 2048 of them up to a million failures), so that several of its
// workgroups stay resident on a CU; the two one-workgroup scans keep 8192.
template <bool ATOMIC, int STAGE>
__device__ __forceinline__ long long block_exclusive_scan_i(const int *src, int m, long long *__restrict__ dst) {
    __shared__ long long wtot[1024 / QSMC_WAVE];
    __shared__ long long carry;
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE, nw = blockDim.x / QSMC_WAVE;
    if (m <= STAGE) {
        // Round 5: ONE sweep.  The loop below takes blockDim entries per trip, each trip a dependent round of loads (agent
        // -scope ones, ~1.5 us, when the counts were written by this very launch) and three barriers: 18 trips for the 4600
        // work items of config 4's share in k_bank_counts' 256-thread workgroup -- most of that kernel's 18 us.  Here every
        // load is issued at once (lane-consecutive: coalesced), the values parked in LDS, and each thread scans a contiguous
        // run of them; integer sums, so the same prefix whatever the order.
        __shared__ int stage[STAGE];
        __syncthreads();
        for (int i = threadIdx.x; i < m; i += (int)blockDim.x)
            stage[i] = ATOMIC ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : src[i];
        __syncthreads();
        const int per = (m + (int)blockDim.x - 1) / (int)blockDim.x;
        const int i0 = (int)threadIdx.x * per;
        long long run = 0ll;
        for (int q = 0; q < per; ++q) run += (i0 + q < m) ? (long long)stage[i0 + q] : 0ll;
        long long inc = run;
#pragma unroll
        for (int off = 1; off < QSMC_WAVE; off <<= 1) {
            const long long t = __shfl_up(inc, off, QSMC_WAVE);
            if (lane >= off) inc += t;
        }
        if (lane == QSMC_WAVE - 1) wtot[wave] = inc;
        __syncthreads();
        long long off0 = inc - run, total = 0ll;
        for (int wv = 0; wv < nw; ++wv) {
            if (wv < wave) off0 += wtot[wv];
            total += wtot[wv];
        }
        for (int q = 0; q < per; ++q) {
            if (i0 + q < m) {
                dst[i0 + q] = off0;
                off0 += (long long)stage[i0 + q];
            }
        }
        if (threadIdx.x == 0) dst[m] = total;
        __syncthreads();
        return total;
    }
    __syncthreads();
    if (threadIdx.x == 0) carry = 0ll;
    __syncthreads();
    for (int base = 0; base < m; base += (int)blockDim.x) {
        const int i = base + (int)threadIdx.x;
        long long v = 0ll;
        if (i < m) v = ATOMIC ? (long long)__hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (long long)src[i];
        long long inc = v;
#pragma unroll
        for (int off = 1; off < QSMC_WAVE; off <<= 1) {
            const long long t = __shfl_up(inc, off, QSMC_WAVE);
            if (lane >= off) inc += t;
        }
        if (lane == QSMC_WAVE - 1) wtot[wave] = inc;
        __syncthreads();
        long long off0 = carry + inc - v;
        for (int wv = 0; wv < wave; ++wv) off0 += wtot[wv];
        if (i < m) dst[i] = off0;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tot = 0ll;
            for (int wv = 0; wv < nw; ++wv) tot += wtot[wv];
            carry += tot;
        }
        __syncthreads();
    }
    const long long total = carry;
    if (threadIdx.x == 0) dst[m] = total;
    __syncthreads();
    return total;
}

// before round 1: prefix of the per-item failure counts, the counters
__global__ __launch_bounds__(1024) void k_bank_scan(BankIn bk, const int *__restrict__ item_off, int chunks) {
    const int n_items = item_off[chunks];
    const long long F = block_exclusive_scan_i<false>(bk.f_cnt, n_items, bk.f_off);
    bank_block_starts(bk.f_off, n_items, bk.vb_first[1]);
    // ([0] = E: k_bank_counts.  Everything else starts from zero -- including the workgroup counters of the NEXT resample's
    //  k_bank_counts and rounds, which is why no memset is queued per resample)
    if (threadIdx.x >= 1 && threadIdx.x < 64) bk.ctr[threadIdx.x] = 0ll;
    __syncthreads();
    if (threadIdx.x == 0) {
        bk.ctr[1] = F;
        bk.ctr[8 + 1] = F;
    }
}

// one failed slot, one spare: true if the spare was invalid too (or the slot was sent to the leftover list)
template <int DM>
__device__ __forceinline__ bool bank_try(const BankIn &bk, long long g, long long E, unsigned int slot, int d,
                                         double *__restrict__ x_out, const OutPlace &pl, bool &left) {
    left = false;
    if (g >= E) {                                     // the bank is exhausted: the old way
        const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long *>(bk.ctr + 2), 1ull);
        bk.leftover[at] = slot;
        left = true;
        return false;
    }
    const long long pg = (long long)bank_perm((unsigned long long)g, (unsigned long long)E, bk.key);
    // (an item's spares start at the prefix of the counts, so spare pg IS entry pg: until round 5 the item was searched for
    //  in e_off -- 13 dependent loads -- only to add e_base[item] - e_off[item] = 0)
    const double *ent = bk.entries + pg * bk.stride;
    if (ent[bk.stride - 1] == 0.0) return true;
    const int64_t row = place_row(pl, (int64_t)slot);
#pragma unroll
    for (int m = 0; m < DM; ++m)
        if (m < d) x_out[m * pl.ld_m + row * pl.ld_s] = ent[m];
    return false;
}

// rounds 1 and 2, over the whole device: the j-th slot still failed (j < F_t) looks at spare perm(B_t + j); the slots
// that fail again are listed per virtual block of 512 (order kept), and the LAST workgroup to finish forms the prefix of
// the block counts for the next round (the counts travel as agent-scope atomics; the slot lists are read by the next
// launch only).
template <int DM>
__global__ __launch_bounds__(BANK_VB) void k_bank_round(BankIn bk, const int *__restrict__ item_off, int chunks, int t, int d,
                                                        const unsigned int *__restrict__ retry_list,
                                                        double *__restrict__ x_out, OutPlace pl) {
    __shared__ int wcount[BANK_VB / QSMC_WAVE];
    __shared__ int is_last;
    __shared__ long long s_win[BANK_VB + 1];
    const long long Ft = bk.ctr[8 + t];
    if (Ft == 0ll) return;
    const long long Bt = bk.ctr[24 + t], E = bk.ctr[0];
    const int n_items = item_off[chunks];
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    const int prev = (t - 1) & 1, cur = t & 1;
    const int nvb_prev = t > 1 ? (int)((bk.ctr[8 + t - 1] + BANK_VB - 1) / BANK_VB) : 0;
    const int nvb = (int)((Ft + BANK_VB - 1) / BANK_VB);
    for (long long vb = blockIdx.x; vb < nvb; vb += gridDim.x) {
        const long long j = vb * BANK_VB + threadIdx.x;
        unsigned int slot = 0u;
        bool fail = false;
        const long long *off = t == 1 ? bk.f_off : bk.boff[prev];
        long long off_src;
        const int src = bank_locate<BANK_VB>(off, t == 1 ? n_items : nvb_prev, bk.vb_first[t & 1][vb], j, j < Ft, s_win, off_src);
        if (j < Ft) {
            if (t == 1) slot = retry_list[bk.f_base[src] + (j - off_src)];
            else slot = bk.tmp[prev][(long long)src * BANK_VB + (j - off_src)];
            bool left;
            fail = bank_try<DM>(bk, Bt + j, E, slot, d, x_out, pl, left);
        }
        const unsigned long long mk = __ballot(fail);
        if (lane == 0) wcount[wave] = __popcll(mk);
        __syncthreads();
        int off0 = 0, tot = 0;
        for (int wv = 0; wv < BANK_VB / QSMC_WAVE; ++wv) {
            if (wv < wave) off0 += wcount[wv];
            tot += wcount[wv];
        }
        if (fail) bk.tmp[cur][vb * BANK_VB + off0 + __popcll(mk & ((1ull << lane) - 1ull))] = slot;
        if (threadIdx.x == 0) __hip_atomic_store(&bk.bcount[cur][vb], tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
    }
    __syncthreads();                                  // (this workgroup's atomic stores have been issued and waited for)
    if (threadIdx.x == 0) {
        const unsigned long long done = __hip_atomic_fetch_add(reinterpret_cast<unsigned long long *>(bk.ctr + 40 + t), 1ull,
                                                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = done == (unsigned long long)gridDim.x - 1ull;
    }
    __syncthreads();
    if (!is_last) return;
    const long long next = block_exclusive_scan_i<true, 2048>(bk.bcount[cur], nvb, bk.boff[cur]);
    bank_block_starts(bk.boff[cur], nvb, bk.vb_first[(t + 1) & 1]);
    if (threadIdx.x == 0) {
        bk.ctr[8 + t + 1] = next;
        bk.ctr[24 + t + 1] = Bt + Ft;
    }
}

// rounds t0 .. BANK_ROUNDS in ONE workgroup (after two rounds a million failures are down to a few thousand: a launch
// per round would be launch latency and nothing else), then whatever is left goes to the leftover list.
template <int DM>
__global__ __launch_bounds__(1024) void k_bank_tail(BankIn bk, const int *__restrict__ item_off, int chunks, int t0, int d,
                                                    double *__restrict__ x_out, OutPlace pl) {
    __shared__ int wcount[1024 / QSMC_WAVE];
    __shared__ long long s_out;
    long long Ft = bk.ctr[8 + t0];
    if (Ft == 0ll) return;
    long long Bt = bk.ctr[24 + t0];
    const long long E = bk.ctr[0];
    __shared__ long long s_win[1024 + 1];
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    const int prev = (t0 - 1) & 1, cur = t0 & 1;
    const int nvb_prev = (int)((bk.ctr[8 + t0 - 1] + BANK_VB - 1) / BANK_VB);
    // this round's slots, densely: out of the previous round's per-block lists
    unsigned int *in = bk.tmp[cur], *out = bk.tmp[cur] + ((Ft + 1023) & ~1023ll);
    for (long long base = 0; base < Ft; base += 1024) {
        const long long j = base + threadIdx.x;
        long long off_pb;
        const int pb = bank_locate<1024>(bk.boff[prev], nvb_prev, bk.vb_first[t0 & 1][base / BANK_VB], j, j < Ft, s_win, off_pb);
        if (j < Ft) in[j] = bk.tmp[prev][(long long)pb * BANK_VB + (j - off_pb)];
        __syncthreads();
    }
    __syncthreads();
    for (int t = t0; t <= BANK_ROUNDS && Ft > 0ll; ++t) {
        if (threadIdx.x == 0) s_out = 0ll;
        __syncthreads();
        for (long long base = 0; base < Ft; base += 1024) {
            const long long j = base + threadIdx.x;
            unsigned int slot = 0u;
            bool fail = false;
            if (j < Ft) {
                slot = in[j];
                bool left;
                fail = bank_try<DM>(bk, Bt + j, E, slot, d, x_out, pl, left);
            }
            const unsigned long long mk = __ballot(fail);
            if (lane == 0) wcount[wave] = __popcll(mk);
            __syncthreads();
            long long off0 = s_out;
            int tot = 0;
            for (int wv = 0; wv < 1024 / QSMC_WAVE; ++wv) {
                if (wv < wave) off0 += wcount[wv];
                tot += wcount[wv];
            }
            if (fail) out[off0 + __popcll(mk & ((1ull << lane) - 1ull))] = slot;
            __syncthreads();
            if (threadIdx.x == 0) s_out += tot;
            __syncthreads();
        }
        Bt += Ft;
        Ft = s_out;
        unsigned int *sw = in;
        in = out;
        out = sw;
        __syncthreads();
    }
    // not served within the queued rounds
    for (long long j = threadIdx.x; j < Ft; j += 1024) {
        const unsigned long long at = atomicAdd(reinterpret_cast<unsigned long long *>(bk.ctr + 2), 1ull);
        bk.leftover[at] = in[j];
    }
}

// ---------------------------------------------------------------------------------------------
// d = 16 (2-qubit tomography) sampler on the matrix cores (k_bucket_anc16 + k_bucket_kick16 below).  The generic kernel's d = 16 instantiation gave every lane
// one output PAIR: 16 Box-Muller pairs and a 2 x 16 x 16 product per lane, 256 VGPRs, two waves per SIMD, 383 us at
// N = 1.25e6 (0.11 of the HBM roofline).  Here a wave owns 16 outputs per trip and the 16 x 16 tile of their kicks
// K = S Z is one v_mfma_f64_16x16x4 chain:
//   lane l = (g = l >> 4, n = l & 15):  B operand of step s = Z[4 g + s][n], the (4 g + s)-th normal of output n --
//     so lane (g, n) draws exactly the two Box-Muller pairs 2 g, 2 g + 1 of output n (blocks o * 8 + 2 g, + 1 of the
//     same Philox stream as before: the normals are the very same numbers);
//   A operand of step s = S[l & 15][4 g + s] (four values per lane, loaded once);
//   D: value r of lane l = K[g + 4 r][n]: each lane ends up with 4 of the 16 coordinates of output n, adds the
//     Liu-West centre of those coordinates (4 gathers) and stores them.
// Position draw and LDS search are done by every lane for its output n (the four lanes of a column agree).
// Only the order in which S z is summed differs from the generic kernel (k = 4 g + s: g inside an MFMA, s across).
// Tomography has no validity constraint (tomography/models.py:143-147), so there is no retry queue here.
// ---------------------------------------------------------------------------------------------
// Gathers: an output needs all 16 coordinates of its ancestor, and in the SoA cloud those sit in 16 rows 8 N bytes
// apart -- 16 cache lines per output, 256 per wave trip when the ancestors of neighbouring outputs are unrelated
// (2.6 GB of line traffic for 160 MB of payload at N = 1.25e6: that, not arithmetic, is what held both d = 16
// kernels at ~370 us).  Outputs are exchangeable, so a work item first HISTOGRAMS its ancestors (one LDS atomic per
// output, after the same position draw and search as before), scans the 4096 counts and expands them into the
// ancestor list in ascending order; trip t then kicks list entries 16 t .. 16 t + 15, whose ancestors are neighbours
// (about one line per row and trip).  Slot o_begin + k takes the k-th smallest ancestor and the normals of slot
// o_begin + k: the same law (the normals are independent of the ancestors), and what oracle/philox.py does too.
typedef double v4d_s __attribute__((ext_vector_type(4)));
constexpr int S16_HEAVY = 24, S16_HEAVY_CAP = BUCKET_CAP / S16_HEAVY + 8;

// Round 3: the d = 16 sampler is TWO kernels.  Phases 1-2 above need only the weights and the plan; the kick needs the
// mean and S = h sqrtm(cov), and at d = 16 the moments behind those are a pass of their own (k_moments_mfma) whose
// result the host turns into S (a 16 x 16 Jacobi, ~20 us).  With one kernel the GPU sat idle through all of that
// (~120 us per resample in round 2).  Split, k_bucket_anc16 is launched the moment the resample is due and runs while
// the host forms S; k_bucket_kick16 is queued behind it as soon as S exists -- before the first has finished.  The
// ancestors travel through HBM as 4-byte indices (5 MB at N = 1.25e6 against 330 MB of particles).  The kick kernel
// no longer carries the chunk tables (70 KB of LDS, two workgroups per CU): 35 KB, five workgroups of four waves.
//
// k_bucket_anc16: anc[o] = source particle (global index) of output slot o; within a work item ascending.
template <int BT>
__global__ __launch_bounds__(BT) void k_bucket_anc16(
    int64_t n_in, const double *__restrict__ w, double inv_norm, const double *__restrict__ offsets, int chunks,
    const long long *__restrict__ slot_off, const int *__restrict__ item_off, const int *__restrict__ item_chunk,
    uint32_t k0, uint32_t k1, uint32_t epoch, unsigned int *__restrict__ anc, int cap,
    unsigned int *__restrict__ canon_count, SqrtJob sq) {
    __shared__ __attribute__((aligned(32))) double lcdf[BUCKET_CHUNK];
    __shared__ double ltops[TOPS_LDS];
    __shared__ unsigned short lguide[TGUIDE_BINS + 2];
    __shared__ double wave_tot[SCAN_WAVES];
    __shared__ int iwave_tot[SCAN_WAVES];
    __shared__ unsigned int cnt2[BUCKET_CHUNK / 2];                 // children per source particle, two counters a word
    __shared__ unsigned int heavy[2 * S16_HEAVY_CAP];
    __shared__ int hcount;
    static_assert(BT == SCAN_THREADS, "one lane owns 8 consecutive source particles");
    static_assert(BUCKET_CAP * 2 <= BUCKET_CHUNK * 8 && BUCKET_CAP < 65536, "the ancestor list overlays the CDF; 16-bit counters");
    // (round 5: 16-bit counters and the list laid over the CDF, as in k_bucket_sample_ordered: 75 -> 51 KB of LDS, three
    //  workgroups per CU instead of two -- config 5's share has ~1500 work items for 256 CUs, i.e. three rounds of them)
    unsigned short *sorted = reinterpret_cast<unsigned short *>(lcdf);  // ancestors of the item's outputs, ascending
    int bid = (int)blockIdx.x;
    if (sq.full) {
        // round 4: workgroup 0 (scheduled first) carries ONE wavefront that turns the summed moments into the Liu-West
        // arguments of the kick kernel (kernels/sqrtm.hpp) while every other workgroup draws ancestors -- the host did this
        // between the two kernels before (~26 us of idle GPU per d = 16 resample)
        if (bid == 0) {
            if (threadIdx.x < QSMC_WAVE) lw_sqrt16_wave(sq, lcdf);
            return;
        }
        bid -= 1;
    }
    // (the kick kernel's list of particles for canonicalize's second pass starts empty: cleared here, a launch earlier,
    //  instead of by a memset command between the two kernels -- that was a 10 us bubble)
    if (bid == 0 && threadIdx.x < 2 && canon_count) canon_count[threadIdx.x] = 0u;      // ([1]: the list pass's second list)
    if (bid >= item_off[chunks]) return;
    const int c = item_chunk[bid];
    const int part = bid - item_off[c];
    const long long slot0 = slot_off[c], n_c = slot_off[c + 1] - slot0;
    const long long t0 = (long long)part * cap;
    const long long t1 = t0 + cap < n_c ? t0 + cap : n_c;
    const int64_t base = (int64_t)c * BUCKET_CHUNK;
    const int len = (int)((n_in - base) < BUCKET_CHUNK ? (n_in - base) : BUCKET_CHUNK);
    const double lo_edge = chunk_edge(offsets, c);
    const double hi_edge = offsets[c + 1];
    const double gscale = (double)TGUIDE_BINS / (hi_edge - lo_edge);
    const bool use_guide = hi_edge > lo_edge && gscale < 1e300;     // workgroup-uniform
    for (int k = threadIdx.x; k < BUCKET_CHUNK / 2; k += BT) cnt2[k] = 0u;
    if (threadIdx.x == 0) hcount = 0;
    chunk_scan_block(w, n_in, inv_norm, offsets, (int64_t)c, wave_tot,
                     StoreLdsTops<0>{lcdf, ltops, lguide, lo_edge, hi_edge, gscale, use_guide, len, -1});
    __syncthreads();
    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;
    const int q = (int)(o_end - o_begin);
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    // ---- 1: ancestors of all q outputs, histogrammed.  Position of slot o: word o & 1 of block (o >> 1, slot 1); a
    // lane takes the pair (2 P, 2 P + 1) so the block is computed once (pairs straddling two items: each its half)
    for (int64_t P = (o_begin >> 1) + threadIdx.x; 2 * P < o_end; P += BT) {
        PhiloxStream rng{(uint64_t)P, (epoch << 16), k0, k1};
        double upos[2];
        rng.uniforms(1, upos[0], upos[1]);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t o = 2 * P + e;
            const double u = lo_edge + upos[e] * (hi_edge - lo_edge);
            int j = table_upper_bound<0>(lcdf, ltops, lguide, (len + 7) >> 3, use_guide, lo_edge, gscale, u);
            j = j > len - 1 ? len - 1 : j;
            if (o >= o_begin && o < o_end) atomicAdd(&cnt2[j >> 1], 1u << (16 * (j & 1)));
        }
    }
    __syncthreads();
    // ---- 2: exclusive scan of the counts (lane l owns particles 8 l .. 8 l + 7), ancestors expanded in ascending order
    // (the list overlays the CDF: the barrier inside, behind the counter reads, is also behind every CDF read above)
    {
        const int j0 = (int)threadIdx.x * 8;
        unsigned int nj[8];
        int lt = 0;
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
            const unsigned int two = cnt2[(j0 + k) >> 1];
            nj[k] = two & 0xffffu;
            nj[k + 1] = two >> 16;
            lt += (int)(nj[k] + nj[k + 1]);
        }
        int inc = lt;
#pragma unroll
        for (int off = 1; off < QSMC_WAVE; off <<= 1) {
            const int t = __shfl_up(inc, off, QSMC_WAVE);
            if (lane >= off) inc += t;
        }
        if (lane == QSMC_WAVE - 1) iwave_tot[wave] = inc;
        __syncthreads();
        int off0 = inc - lt;
        for (int wv = 0; wv < wave; ++wv) off0 += iwave_tot[wv];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int nk = (int)nj[k];
            if (nk > S16_HEAVY) {                                    // a dominant particle: the whole workgroup fills its run
                const int h = atomicAdd(&hcount, 1);
                heavy[2 * h] = (unsigned int)off0 | ((unsigned int)(j0 + k) << 16);
                heavy[2 * h + 1] = (unsigned int)nk;
            } else {
                for (int r = 0; r < nk; ++r) sorted[off0 + r] = (unsigned short)(j0 + k);
            }
            off0 += nk;
        }
        __syncthreads();
        const int nh = hcount;                                      // (uniform: read behind the barrier, not written again)
        if (nh) {
            for (int h = 0; h < nh; ++h) {
                const unsigned int st = heavy[2 * h] & 0xffffu, jj = heavy[2 * h] >> 16, cn = heavy[2 * h + 1];
                for (unsigned int r = threadIdx.x; r < cn; r += BT) sorted[st + r] = (unsigned short)jj;
            }
            __syncthreads();
        }
    }
    for (int k = threadIdx.x; k < q; k += BT) anc[o_begin + k] = (unsigned int)(base + (int64_t)sorted[k]);
}

// k_bucket_kick16: the kicks of all output slots, 64 per wave trip, and -- CANON != 0 -- the first pass of
// TomographyModel.canonicalize on the way out (smc.py:529, tomography/models.py:149-209).
// Per sub-trip t = 0..3 a wave does what one trip of the round-2 kernel did for 16 slots: lane (g, n) gathers 4
// coordinates of the ancestor of slot k0 + 16 t + n, draws its two Box-Muller pairs and the 16 x 16 tile of kicks
// K = S Z is one v_mfma_f64_16x16x4 chain (operand layout as described above); the results go to an LDS tile
// [slot][coordinate] (row stride 17: the transposed read below is conflict-free up to 2 ways).  After four
// sub-trips lane L owns ALL 16 coordinates of slot k0 + L:
//   * CANON: the LDL^H pivot test (tomo_clearly_positive); a positive-definite rho (two thirds of a fresh cloud) is
//     finished here (x / (x_0 sqrt dim)), the others are listed for k_tomo_canon_list -- round 2 ran this test as a
//     separate pass (k_tomo_classify) that read the cloud back and wrote it again: 320 MB and 61 us per resample;
//   * the particle is written once, row by row: 64 consecutive slots of one coordinate per store instruction (512 B),
//     where the MFMA layout's own stores were 16 slots x 4 coordinates (128 B segments; 1.27x the algorithmic bytes).
// Same arithmetic per particle as before the split: (a x_a + (1 - a) mu) + (S z), then p * inv -- bit-identical clouds.
// Workgroup b takes slots [r per_block, (r + 1) per_block), r = (b & 7) per + (b >> 3): ranges adjacent in slot order
// (whose ancestors are neighbours) go to ONE XCD (b % 8) back to back, so each 16 x 32 KB source window is fetched by
// one L2 instead of eight.
// Round 6 built the "bytes lever" the round-5 counters pointed at (212.7 MB of read requests for 165 MB of payload: a chunk's
// work items each stream through the chunk's whole source window): workgroups aligned to work items, each taking the same
// FRACTION of every item of its chunk, four waves side by side on the same source lines.  Read requests fell to 128.9 MB
// (TCC_EA0_RDREQ x 64 B; below the payload, because lines without a child are never wanted) -- and the kernel went from 101
// to 115 us on the same box, as did plain item-aligned workgroups (115) while the same loop over fixed 1024-slot ranges
// stayed at 104: workgroups that start at the start of their chunk's window walk it in a convoy and wait on the same
// lines in flight; ranges that straddle items are spread over the window.  The re-read lines come out of the Infinity Cache,
// not HBM, and do not bound this kernel (VALU 54 % busy, gathers).  Not adopted: tools/experiments/
// r6_kick16_fractions_of_every_item.patch, profiles/r6_g_kick16_modes.txt.
constexpr int KICK16_BT = 256, KICK16_WAVES = KICK16_BT / QSMC_WAVE, KICK16_ROW = 17, KICK16_PER_BLOCK = 1024;
template <int CANON>       // 0: no canonicalize; 1: the 2-qubit Pauli basis (sparse contraction); 2: dense basis
__global__ __launch_bounds__(KICK16_BT) void k_bucket_kick16(
    const double *__restrict__ x_in, int64_t ldx_in, const unsigned int *__restrict__ anc, int64_t n_out, LWArgs lw,
    uint32_t k0, uint32_t k1, uint32_t epoch, double *__restrict__ x_out, OutPlace pl,
    const double *__restrict__ basis, int allow_subnormalized, unsigned int *__restrict__ list,
    unsigned int *__restrict__ count, const LWDev *__restrict__ lwd) {
    constexpr int DM = 16;
    // lwd != nullptr: a, mean and S come from device memory (written by lw_sqrt16_wave in the kernel before this one);
    // a block marked invalid (covariance or square-root error not finite) means the host will not adopt this resample
    if (lwd && lwd->valid != 1.0) return;
    const double lw_a = lwd ? lwd->a : lw.a;
    __shared__ double tile[KICK16_WAVES][QSMC_WAVE * KICK16_ROW];
    __shared__ double sS[DM * DM + DM];                             // S (row-major) and the mean
    __shared__ unsigned int hard_buf[KICK16_PER_BLOCK];
    __shared__ unsigned int bcount, gbase;
    const int64_t n_ranges = (n_out + KICK16_PER_BLOCK - 1) / KICK16_PER_BLOCK;
    const int per = ((int)gridDim.x + 7) >> 3;
    const int64_t rb = (int64_t)((int)blockIdx.x & 7) * per + ((int)blockIdx.x >> 3);
    if (((int)blockIdx.x >> 3) >= per || rb >= n_ranges) return;
    const int64_t r0 = rb * KICK16_PER_BLOCK;
    const int64_t r1 = r0 + KICK16_PER_BLOCK < n_out ? r0 + KICK16_PER_BLOCK : n_out;
    for (int k = threadIdx.x; k < DM * DM; k += KICK16_BT) sS[k] = lwd ? lwd->S[k] : lw.S[k];
    if (threadIdx.x < DM) sS[DM * DM + threadIdx.x] = lwd ? lwd->mean[threadIdx.x] : lw.mean[threadIdx.x];
    if (threadIdx.x == 0) bcount = 0u;
    __syncthreads();
    const int lane = threadIdx.x & (QSMC_WAVE - 1), wave = threadIdx.x / QSMC_WAVE;
    const int n = lane & 15, g = lane >> 4;
    double aS[4], mu4[4];
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) aS[sidx] = sS[(lane & 15) * DM + 4 * g + sidx];
#pragma unroll
    for (int r = 0; r < 4; ++r) mu4[r] = (1.0 - lw_a) * sS[DM * DM + g + 4 * r];
    double *mine = tile[wave];
    for (int64_t kb = r0; kb < r1; kb += KICK16_BT) {               // (uniform over the workgroup)
        const int64_t k64 = kb + (int64_t)wave * QSMC_WAVE;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int64_t k = k64 + 16 * t + n;
            const int64_t oc = k < r1 ? k : r1 - 1;                 // (idle columns shadow the last slot: no divergence)
            const int64_t j = (int64_t)anc[oc];
            double xa[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) xa[r] = x_in[(int64_t)(g + 4 * r) * ldx_in + j];
            // the four normals 4 g .. 4 g + 3 of slot oc: pairs 2 g and 2 g + 1 (blocks oc * 8 + 2 g, + 1; slot 2)
            double z[4];
            PhiloxStream nrm{(uint64_t)oc * 8u + (uint64_t)(2 * g), (epoch << 16), k0, k1};
            nrm.normals(2, z[0], z[1]);
            nrm.particle += 1;
            nrm.normals(2, z[2], z[3]);
            v4d_s acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[sidx], z[sidx], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) mine[(16 * t + n) * KICK16_ROW + g + 4 * r] = (lw_a * xa[r] + mu4[r]) + acc[r];
        }
        // (the tile is this WAVE's: a wave-level fence orders its LDS stores before its own transposed reads.  Round 5 --
        //  the two workgroup barriers per trip that stood here kept a workgroup's four waves in lockstep through a kernel
        //  whose waves stall on gathers half the time: VALUs 54 % busy at three waves per SIMD)
        wave_lds_sync();
        const int64_t o = k64 + lane;
        bool hard = false;
        if (o < r1) {
            double p[DM];
#pragma unroll
            for (int m = 0; m < DM; ++m) p[m] = mine[lane * KICK16_ROW + m];
            if (CANON) {
                bool pos;
                if (CANON == 1) pos = tomo_clearly_positive<4>(TomoPauli2{}, p);
                else pos = tomo_clearly_positive<4>(TomoDense<4>{basis}, p);
                if (pos) {
                    if (!allow_subnormalized) {                     // tomography/models.py:194-209
                        const double inv = 1.0 / (p[0] * sqrt(4.0));
#pragma unroll
                        for (int m = 0; m < DM; ++m) p[m] = p[m] * inv;
                    }
                } else {
                    hard = true;
                }
            }
            const int64_t row = place_row(pl, o);
#pragma unroll
            for (int m = 0; m < DM; ++m) x_out[(int64_t)m * pl.ld_m + row * pl.ld_s] = p[m];
        }
        if (CANON) {
            const unsigned long long mk = __ballot(hard);
            if (mk) {
                unsigned int hb = 0;
                if (lane == 0) hb = atomicAdd(&bcount, (unsigned int)__popcll(mk));
                hb = __shfl(hb, 0, QSMC_WAVE);
                if (hard) hard_buf[hb + __popcll(mk & ((1ull << lane) - 1ull))] = (unsigned int)o;
            }
        }
        wave_lds_sync();                                            // the wave's tile is rewritten by its next trip
    }
    if (CANON) {
        __syncthreads();                                            // every wave's entries of hard_buf and its share of bcount
        const unsigned int m = bcount;
        if (m) {
            if (threadIdx.x == 0) gbase = atomicAdd(count, m);
            __syncthreads();
            for (unsigned int t = threadIdx.x; t < m; t += KICK16_BT) list[gbase + t] = hard_buf[t];
        }
    }
}

// Second chance for the queued outputs: redraw ancestor and kick from the global CDF (rounds 1..).  One launch,
// resident grid: nothing queued (most resamples) -> leave at once (an empty launch is ~5 us;
// the former pair -- materialise the CDF behind a gate, then redraw -- was two of them).  Otherwise every
// workgroup scans its share of the chunks into the global CDF, all meet at a barrier, and the queue is worked off.
// Held to 128 VGPRs (4 waves per SIMD): two workgroups fit a CU, so 256 are resident on half the CUs and two
// processes sharing a GPU (as the tests do) both stay resident; the scan phase takes ~10 rounds instead of 19.
constexpr int REDRAW_BLOCKS = 256;
constexpr int REDRAW_SMALL_MAX = 8192;      // queued outputs up to which the redraw works chunk by chunk in LDS (no global CDF)
template <int DM>     // particle dimension bound: 4 (registers) or QSMC_MAX_D
__attribute__((amdgpu_waves_per_eu(4, 8)))
__global__ __launch_bounds__(SCAN_THREADS) void k_bucket_redraw(
    int kind, int d, double min_freq, const double *__restrict__ x_in, int64_t ldx_in, int64_t n_in,
    const double *__restrict__ w, double inv_norm, const double *__restrict__ offsets, int64_t chunks, double *cdf,
    LWArgs lw, uint32_t k0, uint32_t k1, uint32_t epoch, int maxiter, double *__restrict__ x_out, OutPlace pl,
    const unsigned int *__restrict__ retry_list, const unsigned long long *__restrict__ retry_count,
    unsigned long long *__restrict__ n_failed, unsigned long long *bar, int cdf_ready, int edges_in_lds, int small_lds) {
    __shared__ double wave_tot[SCAN_WAVES];
    extern __shared__ __attribute__((aligned(16))) unsigned char redraw_smem[];
    const unsigned long long cnt = *retry_count;
    if (cnt == 0ull) return;
    if (!cdf_ready && cnt <= (unsigned long long)REDRAW_SMALL_MAX && small_lds) {
        // Round 4: a handful of queued outputs (precession's omega > 0 bites at a few early resamples: 2-3 % of config 2's,
        // one in six of config 3's) used to cost the whole one-launch form -- every workgroup scanning its share of ALL
        // chunks into the global CDF, 160 MB of traffic and a grid barrier, ~150 us, for a few hundred particles.  Here a
        // workgroup takes one queued output at a time: the chunk its draw falls into (binary search of the offsets) is
        // scanned into LDS by the same routine that fills the global CDF (chunk_scan_block: the same numbers), searched,
        // kicked, tested; next round if invalid.  Same Philox blocks, same CDF entries, same particles as the global form.
        double *lcdf = reinterpret_cast<double *>(redraw_smem);
        __shared__ int s_j, s_ok;
        unsigned long long failed_small = 0;
        for (unsigned long long i = blockIdx.x; i < cnt; i += gridDim.x) {
            const int64_t o = (int64_t)retry_list[i];
            double p[DM];
            bool ok = false;
            for (int round = 1; round < maxiter && !ok; ++round) {          // (uniform over the workgroup)
                PhiloxStream rng{(uint64_t)o, (epoch << 16) | (uint32_t)round, k0, k1};
                double u0, unused;
                rng.uniforms(0, u0, unused);
                int64_t lo = 0, hi = chunks;                                 // #upper edges <= u0 = the chunk of the upper bound
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (offsets[mid + 1] <= u0) lo = mid + 1; else hi = mid;
                }
                const int64_t c = lo > chunks - 1 ? chunks - 1 : lo;
                __syncthreads();                                             // (lcdf / s_ok of the previous round are done with)
                chunk_scan_block(w, n_in, inv_norm, offsets, c, wave_tot, StoreGlobal{lcdf});
                __syncthreads();
                if (threadIdx.x == 0) {
                    const int64_t base = c * SCAN_CHUNK;
                    const int64_t len = n_in - base < SCAN_CHUNK ? n_in - base : SCAN_CHUNK;
                    int64_t l2 = 0, h2 = len;
                    while (l2 < h2) {
                        const int64_t mid = (l2 + h2) >> 1;
                        if (lcdf[mid] <= u0) l2 = mid + 1; else h2 = mid;
                    }
                    const int64_t j = base + l2 < n_in - 1 ? base + l2 : n_in - 1;
                    double zz[DM];
#pragma unroll
                    for (int q = 0; q < DM; q += 2) {
                        if (q < d) {
                            double z0, z1;
                            rng.normals(1 + (q >> 1), z0, z1);
                            zz[q] = z0;
                            if (q + 1 < DM) zz[q + 1] = z1;
                        }
                    }
#pragma unroll
                    for (int m = 0; m < DM; ++m) {
                        if (m < d) {
                            double sm = 0.0;
#pragma unroll
                            for (int q = 0; q < DM; ++q)
                                if (q < d) sm += lw.S[m * d + q] * zz[q];
                            p[m] = (lw.a * x_in[m * ldx_in + j] + (1.0 - lw.a) * lw.mean[m]) + sm;
                        }
                    }
                    s_ok = model_valid(kind, p, min_freq) ? 1 : 0;
                }
                __syncthreads();
                ok = s_ok != 0;
            }
            if (threadIdx.x == 0) {
                const int64_t row = place_row(pl, o);      // like the global form: the last round's value stays
                for (int m = 0; m < d; ++m) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
                if (!ok) ++failed_small;
            }
        }
        if (threadIdx.x == 0 && failed_small) atomicAdd(n_failed, failed_small);
        return;
    }
    double *edges = edges_in_lds ? reinterpret_cast<double *>(redraw_smem) : nullptr;
    if (edges) {                                         // upper edge of every chunk (offsets[c + 1]); read after the
        for (int c = threadIdx.x; c < (int)chunks; c += SCAN_THREADS) edges[lds_skew(c)] = offsets[c + 1];   // scans' barriers
    }
    if (!cdf_ready) {                                                // (cdf_ready: a full-grid k_chunk_scan ran before this launch)
        for (int64_t c = blockIdx.x; c < chunks; c += gridDim.x) {
            chunk_scan_block(w, n_in, inv_norm, offsets, c, wave_tot, StoreGlobal{cdf + c * SCAN_CHUNK});
            __syncthreads();                                         // wave_tot is reused by the next chunk
        }
        grid_barrier_fenced(bar);
    }
    __syncthreads();
    unsigned long long failed = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * SCAN_THREADS + threadIdx.x; i < cnt;
         i += (unsigned long long)gridDim.x * SCAN_THREADS) {
        const int64_t o = (int64_t)retry_list[i];
        double p[DM];
        const bool ok = redraw_rounds<DM>(kind, d, min_freq, x_in, ldx_in, n_in, cdf, lw, k0, k1, epoch,
                                                  maxiter, o, p, edges, (int)chunks);
        const int64_t row = place_row(pl, o);      // like the in-thread loop: the last round's value stays
        for (int m = 0; m < d; ++m) x_out[m * pl.ld_m + row * pl.ld_s] = p[m];
        if (!ok) ++failed;
    }
    if (failed) atomicAdd(n_failed, failed);
}

// copies the failed-particle counter into pinned host memory (read later, after any stream sync)
__global__ void k_publish_counter(const unsigned long long *__restrict__ counter, double *__restrict__ mapped_slot) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        mapped_slot[0] = (double)counter[0];
        mapped_slot[-1] = (double)counter[1];          // how many outputs of that resample needed a global redraw
    }
}

__global__ __launch_bounds__(QSMC_BLOCK) void k_prior_uniform_philox(
    int kind, int d, double min_freq, int postselect, LWArgs box /* mean = lo, S[0..d) = hi - lo */,
    int64_t n, uint32_t k0, uint32_t k1, uint32_t epoch, int maxiter, double *__restrict__ x_out,
    int64_t ldx_out, unsigned long long *__restrict__ n_failed) {
    unsigned long long failed = 0;
    for (int64_t i = (int64_t)blockIdx.x * QSMC_BLOCK + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * QSMC_BLOCK) {
        double p[QSMC_MAX_D];
        bool ok = false;
        for (int round = 0; round < maxiter && !ok; ++round) {
            PhiloxStream rng{(uint64_t)i, (epoch << 16) | (uint32_t)round, k0, k1};
            for (int q = 0; q < d; q += 2) {
                double u0, u1;
                rng.uniforms(q >> 1, u0, u1);
                p[q] = box.mean[q] + u0 * box.S[q];                         // lo + z * delta (:818-819)
                if (q + 1 < d) p[q + 1] = box.mean[q + 1] + u1 * box.S[q + 1];
            }
            ok = !postselect || model_valid(kind, p, min_freq);
        }
        for (int m = 0; m < d; ++m) x_out[m * ldx_out + i] = p[m];
        if (!ok) ++failed;
    }
    if (failed) atomicAdd(n_failed, failed);
}

