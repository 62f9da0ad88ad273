// What a FLAT d = 3 kick kernel would cost (measurement tooling): the third phase of
// k_bucket_sample_ordered<3> as a kernel of its own -- ancestors read as 4-byte indices (ascending inside runs of 4096
// slots, like the ordered sampler's items), three Box-Muller blocks per output pair, the Liu-West combine, RB's validity
// test, a failure bit per slot, SoA stores -- with no chunk tables in LDS, so occupancy is set by registers alone.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I python-qinfer_amd/csrc tools/lat/kick3.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "qsmc_device.h"
using namespace qsmc;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
struct KArgs { double a, mean[3], S[9]; };
template <int BT>
__global__ __launch_bounds__(BT) void k_kick3(const double *__restrict__ x_in, int64_t ld_in, const unsigned int *__restrict__ anc,
                                             int64_t n_out, KArgs lw, uint32_t k0, uint32_t k1, uint32_t epoch,
                                             double *__restrict__ x_out, int64_t ld_out, unsigned int *__restrict__ failbits) {
    const int64_t P = (int64_t)blockIdx.x * BT + threadIdx.x;
    if (2 * P >= n_out) return;
    double xg[2][3];
    int64_t o[2] = {2 * P, 2 * P + 1 < n_out ? 2 * P + 1 : 2 * P};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const unsigned int j = anc[o[e]];
#pragma unroll
        for (int m = 0; m < 3; ++m) xg[e][m] = x_in[m * ld_in + j];
    }
    double z[6];
    PhiloxStream nrm{0, (epoch << 16), k0, k1};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        nrm.particle = (uint64_t)P * 3ull + (uint64_t)k;
        nrm.normals(2, z[2 * k], z[2 * k + 1]);
    }
    unsigned int bad = 0u;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        double p[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            double sm = 0.0;
#pragma unroll
            for (int q = 0; q < 3; ++q) sm += lw.S[m * 3 + q] * z[e * 3 + q];
            p[m] = (lw.a * xg[e][m] + (1.0 - lw.a) * lw.mean[m]) + sm;
        }
        if (!Model<QSMC_MODEL_RB>::valid(p, 0.0)) bad |= 1u << e;
        if (e == 0 || 2 * P + 1 < n_out) {
#pragma unroll
            for (int m = 0; m < 3; ++m) x_out[m * ld_out + o[e]] = p[m];
        }
    }
    // two bits per lane -> two 64-bit ballots -> four words per wave
    const unsigned long long b0 = __ballot(bad & 1u), b1 = __ballot(bad & 2u);
    if ((threadIdx.x & 63) == 0) {
        unsigned int *w = failbits + ((2 * P) >> 5);
        w[0] = (unsigned int)b0; w[1] = (unsigned int)(b0 >> 32); w[2] = (unsigned int)b1; w[3] = (unsigned int)(b1 >> 32);
    }
}
int main() {
    const int64_t n = 12500000;
    double *x_in, *x_out; unsigned int *anc, *fb;
    CK(hipMalloc(&x_in, 3 * n * 8)); CK(hipMalloc(&x_out, 3 * n * 8)); CK(hipMalloc(&anc, n * 4)); CK(hipMalloc(&fb, n / 8 + 64));
    std::vector<double> hx(3 * n);
    for (int64_t i = 0; i < n; ++i) { hx[i] = 0.9 + 0.1 * (i % 997) / 997.0; hx[n + i] = 0.2 + 0.3 * (i % 991) / 991.0; hx[2 * n + i] = 0.4 + 0.2 * (i % 983) / 983.0; }
    CK(hipMemcpy(x_in, hx.data(), 3 * n * 8, hipMemcpyHostToDevice));
    // ancestors: within each run of 4096 slots ascending indices of the same 4096-particle chunk, ~0.6 children per particle
    std::vector<unsigned int> ha(n);
    uint64_t s = 12345;
    for (int64_t b = 0; b < n; b += 4096) {
        int64_t m = n - b < 4096 ? n - b : 4096, j = 0;
        for (int64_t k = 0; k < m; ++k) {
            s = s * 6364136223846793005ull + 1442695040888963407ull;
            j += ((s >> 33) % 3 == 0) ? 0 : ((s >> 40) & 1) + 1;     // steps of 0, 1, 2
            ha[b + k] = (unsigned int)(b + (j < m ? j : m - 1));
        }
    }
    CK(hipMemcpy(anc, ha.data(), n * 4, hipMemcpyHostToDevice));
    KArgs lw; lw.a = 0.98; lw.mean[0] = 0.95; lw.mean[1] = 0.35; lw.mean[2] = 0.5;
    for (int i = 0; i < 9; ++i) lw.S[i] = (i % 4 == 0) ? 0.004 : 0.0005;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](auto kern, int bt, const char *name) {
        const int grid = (int)((n / 2 + bt - 1) / bt);
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, grid, bt, 0, 0, x_in, n, anc, n, lw, 1u, 2u, 3u, x_out, n, fb);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid, bt, 0, 0, x_in, n, anc, n, lw, 1u, 2u, 3u, x_out, n, fb);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.1f us per launch (N = %ld, d = 3; 650 MB: %.2f TB/s)\n", name, ms * 50.0, (long)n, 650e6 / (ms * 50.0e-6) / 1e12);
    };
    run(k_kick3<256>, 256, "flat kick, 256 threads");
    run(k_kick3<512>, 512, "flat kick, 512 threads");
    return 0;
}
