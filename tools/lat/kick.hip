// Issue-rate anatomy of the Liu-West kick: Philox4x32-10 -> two 53-bit uniforms -> Box-Muller pair, as the sampler
// and k_random_walk run it.  Variants isolate the parts; each thread does PAIRS_PER_THREAD pairs in a grid-stride loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../python-qinfer_amd/csrc kick.hip -o kick.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include "qsmc_device.h"
using namespace qsmc;

template <int V, int WAVES_MIN>
__attribute__((amdgpu_waves_per_eu(WAVES_MIN, 8)))
__global__ __launch_bounds__(256) void k(double *__restrict__ out, int64_t n_pairs, uint32_t k0, uint32_t k1) {
    for (int64_t P = (int64_t)blockIdx.x * 256 + threadIdx.x; P < n_pairs; P += (int64_t)gridDim.x * 256) {
        double a, b;
        if (V == 0) {                       // Philox + u53 only
            PhiloxStream r{(uint64_t)P, 1u << 16, k0, k1};
            r.uniforms(2, a, b);
        } else if (V == 1) {                // Philox + Box-Muller (the kick's generator)
            PhiloxStream r{(uint64_t)P, 1u << 16, k0, k1};
            r.normals(2, a, b);
        } else if (V == 2) {                // Box-Muller only, uniforms from a cheap hash
            const uint32_t h = (uint32_t)P * 2654435761u;
            const double u0 = (double)(h >> 8) * (1.0 / 16777216.0), u1 = (double)((h * 40503u) >> 8) * (1.0 / 16777216.0);
            const double rr = bm_sqrt(-2.0 * bm_log(1.0 - u0 * 0.999));
            double s, c;
            bm_sincospi(2.0 * u1, s, c);
            a = rr * c; b = rr * s;
        } else if (V == 3) {                // log + sqrt only
            const uint32_t h = (uint32_t)P * 2654435761u;
            const double u0 = (double)(h >> 8) * (1.0 / 16777216.0);
            a = bm_sqrt(-2.0 * bm_log(1.0 - u0 * 0.999)); b = a;
        } else {                            // sincospi only
            const uint32_t h = (uint32_t)P * 2654435761u;
            const double u1 = (double)(h >> 8) * (1.0 / 16777216.0);
            bm_sincospi(2.0 * u1, a, b);
        }
        *reinterpret_cast<double2 *>(out + 2 * P) = double2{a, b};
    }
}

template <int V, int W>
static void run(const char *name, double *out, int64_t n_pairs) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<V, W>), dim3(grid), dim3(256), 0, 0, out, n_pairs, 1u, 2u);
    hipEventRecord(e0);
    for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((k<V, W>), dim3(grid), dim3(256), 0, 0, out, n_pairs, 1u, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s waves>=%d  %7.1f us per 5e6 pairs (1e7 normals)\n", name, W, ms * 100.0);
}

int main() {
    const int64_t n_pairs = 5000000;
    double *out;
    hipMalloc(&out, n_pairs * 16);
    run<0, 1>("philox + 2 x u53", out, n_pairs);
    run<1, 1>("philox + box-muller", out, n_pairs);
    run<1, 8>("philox + box-muller", out, n_pairs);
    run<2, 1>("box-muller only", out, n_pairs);
    run<3, 1>("log + sqrt only", out, n_pairs);
    run<4, 1>("sincospi only", out, n_pairs);
    return 0;
}
