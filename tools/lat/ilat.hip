// Dependent-issue latency microbenchmark for gfx950 (measurement tooling): one wave per SIMD, ILP = 1, 2, 4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
constexpr int ITER = 8192;
template <int OP, int ILP>
__global__ __launch_bounds__(64) void k(double *out, double seed, long long *cyc) {
    double a[ILP]; uint64_t u[ILP]; uint32_t w[ILP];
    for (int i = 0; i < ILP; ++i) { a[i] = seed + i * 0.001 + threadIdx.x * 1e-6; u[i] = (uint64_t)(a[i] * 1e9); w[i] = (uint32_t)u[i] | 1u; }
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) a[i] = fma(a[i], 1.0000001, 1e-9);
            if (OP == 1) u[i] = (uint64_t)(uint32_t)u[i] * 0xD2511F53ull + (u[i] >> 32);
            if (OP == 2) w[i] = (w[i] ^ 0x1234567u) + (w[i] >> 3);
            if (OP == 3) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == 4) a[i] = a[i] + 1e-9;
        }
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < ILP; ++i) s += a[i] + (double)u[i] + (double)w[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP, int ILP> void run(const char *name) {
    double *out; long long *cyc, hc;
    hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 8);
    hipLaunchKernelGGL((k<OP, ILP>), 1, 64, 0, 0, out, 1.0, cyc);
    hipLaunchKernelGGL((k<OP, ILP>), 1, 64, 0, 0, out, 1.0, cyc);
    hipDeviceSynchronize();
    hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s ILP %d: %.2f clock64 ticks per instruction (single wave)\n", name, ILP, (double)hc / (ITER * ILP));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 1>("v_fma_f64"); run<0, 2>("v_fma_f64"); run<0, 4>("v_fma_f64"); run<0, 8>("v_fma_f64");
    run<4, 1>("v_add_f64"); run<4, 2>("v_add_f64");
    run<1, 1>("v_mad_u64_u32"); run<1, 2>("v_mad_u64_u32"); run<1, 4>("v_mad_u64_u32");
    run<2, 1>("int32 xor/shift/add (3)"); run<2, 2>("int32 xor/shift/add (3)");
    run<3, 1>("v_rcp_f64"); run<3, 2>("v_rcp_f64");
    return 0;
}
