// Instruction issue-rate microbenchmark for gfx950 (measurement tooling): cycles per wave64 instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s\n", hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITER = 4096, ILP = 8;
template <int OP>
__global__ __launch_bounds__(256) void k(double *out, double seed, long long *cyc) {
    double a[ILP]; uint64_t u[ILP]; uint32_t w[ILP];
    for (int i = 0; i < ILP; ++i) { a[i] = seed + i * 0.001 + threadIdx.x * 1e-6; u[i] = (uint64_t)(a[i] * 1e9); w[i] = (uint32_t)u[i] | 1u; }
    long long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) a[i] = fma(a[i], 1.0000001, 1e-9);
            if (OP == 1) a[i] = a[i] * 1.0000001;
            if (OP == 2) a[i] = a[i] + 1e-9;
            if (OP == 3) u[i] = (uint64_t)(uint32_t)u[i] * 0xD2511F53ull + (u[i] >> 32);       // v_mad_u64_u32
            if (OP == 4) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == 5) a[i] = __builtin_amdgcn_rsq(a[i]);
            if (OP == 6) w[i] = w[i] * 0x9E3779B9u + 1u;                                          // v_mul_lo_u32 (+add)
            if (OP == 7) w[i] = __umulhi(w[i], 0xD2511F53u) ^ 0x55u;
            if (OP == 8) a[i] = rint(a[i] * 1.5);
            if (OP == 9) w[i] = (w[i] ^ (w[i] >> 7)) + 0x1234567u;                                // xor/shift/add int32
            if (OP == 10) a[i] = __builtin_amdgcn_sqrt(a[i]);
            if (OP == 11) a[i] = (double)(int)(a[i]) + 1.5;                                         // cvt i32<->f64
            if (OP == 12) u[i] = u[i] + 0x9E3779B97F4A7C15ull;                                      // 64-bit add
            if (OP == 13) a[i] = fmin(fmax(a[i], 0.5), 2.0);
            if (OP == 14) {     // u53 as the library forms it: two shifts, two v_cvt_f64_u32, fma, mul  (+ an xor to chain)
                const uint32_t x = w[i], y = w[i] * 3u + (uint32_t)it;
                a[i] = ((double)(x >> 5) * 67108864.0 + (double)(y >> 6)) / 9007199254740992.0;
                w[i] = x ^ (uint32_t)__double2loint(a[i]);
            }
            if (OP == 15) {     // the same value from bits: 0.5 + low 52 bits at exponent -1, minus 0.5 unless bit 52 is set
                const uint32_t x = w[i], y = w[i] * 3u + (uint32_t)it;
                const uint32_t hi = 0x3FE00000u | ((x >> 11) & 0xFFFFFu), lo = ((x << 21) & 0xFC000000u) | (y >> 6);
                const double v = __hiloint2double((int)hi, (int)lo);
                a[i] = v - ((int)x < 0 ? 0.0 : 0.5);
                w[i] = x ^ (uint32_t)__double2loint(a[i]);
            }
        }
    }
    long long t1 = clock64();
    double s = 0; for (int i = 0; i < ILP; ++i) s += a[i] + (double)u[i] + (double)w[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> int run(const char *name, int waves_per_simd) {
    double *out; long long *cyc, hc;
    CHECK(hipMalloc(&out, 4096 * 256 * 8)); CHECK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves_per_simd;       // 256 CUs, 4 waves per block = 1 wave per SIMD per block
    hipLaunchKernelGGL(k<OP>, blocks, 256, 0, 0, out, 1.0, cyc);
    hipEventRecord(e0); hipLaunchKernelGGL(k<OP>, blocks, 256, 0, 0, out, 1.0, cyc); hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms; hipEventElapsedTime(&ms, e0, e1);
    CHECK(hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost));
    // per-SIMD instruction count = waves_per_simd * ITER * ILP
    double n = (double)waves_per_simd * ITER * ILP;
    printf("%-28s waves/SIMD %d: %.2f ns per wave-instr per SIMD  (clock64 delta/instr %.2f)\n", name, waves_per_simd,
           ms * 1e6 / n, (double)hc / (ITER * ILP));
    hipFree(out); hipFree(cyc);
    return 0;
}
int main() {
    for (int wps : {1, 4}) {
        run<0>("v_fma_f64", wps); run<1>("v_mul_f64", wps); run<2>("v_add_f64", wps); run<3>("v_mad_u64_u32", wps);
        run<4>("v_rcp_f64", wps); run<5>("v_rsq_f64", wps); run<6>("v_mul_lo_u32+add", wps); run<7>("v_mul_hi_u32+xor", wps);
        run<8>("mul_f64+rndne_f64", wps); run<9>("xor/shift/add i32 (3 ops)", wps); run<10>("v_sqrt_f64", wps);
        run<11>("cvt f64->i32->f64 + add", wps); run<12>("u64 add (2 ops)", wps); run<13>("fmax+fmin f64", wps);
        run<14>("u53 via 2 cvt (+mul,xor chain)", wps); run<15>("u53 via bits (+mul,xor chain)", wps);
    }
    return 0;
}
