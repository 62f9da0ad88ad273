"""Build ablation variants of libqsmc_hip.so (tools/abl_libs/libqsmc_abl<N>.so) from a patched temp copy
of the kernel source.  Measurement tooling only; nothing here ships."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csrc = os.path.join(root, 'python-qinfer_amd/csrc')
src = open(os.path.join(csrc, 'qsmc_kernels.hip')).read()
for line in [l for l in src.split('\n') if l.startswith('#include "kernels/')]:     # flatten: the anchors below span files
    part = open(os.path.join(csrc, line.split('"')[1])).read().replace('#pragma once\n', '')
    src = src.replace(line, part, 1)
def rep(s, old, new):
    assert old in s, old[:60]
    return s.replace(old, new, 1)
s = src
s = rep(s, "    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;         // this item's output slots\n    unsigned long long failed = 0;",
 "    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;         // this item's output slots\n    unsigned long long failed = 0;\n#if QSMC_ABL == 1 || QSMC_ABL == 7 || QSMC_ABL == 8\n    if (lguide[3] != 12345 || lcdf[5] != 0.123) return;\n#endif")
s = rep(s, "    const bool use_guide = hi_edge > lo_edge && gscale < 1e300;     // workgroup-uniform\n    chunk_scan_block(",
 "#if QSMC_ABL == 7 || QSMC_ABL == 8\n    const bool use_guide = false;\n#else\n    const bool use_guide = hi_edge > lo_edge && gscale < 1e300;\n#endif\n    chunk_scan_block(")
s = rep(s, "        const double inc = wave_inclusive_scan(v[SCAN_PER_LANE - 1], lane);",
 "#if QSMC_ABL == 8\n        const double inc = v[SCAN_PER_LANE - 1];\n#else\n        const double inc = wave_inclusive_scan(v[SCAN_PER_LANE - 1], lane);\n#endif")
s = rep(s, "    const double m = wave_inclusive_max(v[SCAN_PER_LANE - 1], lane);",
 "#if QSMC_ABL == 8\n    const double m = v[SCAN_PER_LANE - 1];\n#else\n    const double m = wave_inclusive_max(v[SCAN_PER_LANE - 1], lane);\n#endif")
s = rep(s, "__attribute__((amdgpu_waves_per_eu(D >= 1 && D <= 2 ? 6 : 1, 8)))",
 "#if QSMC_ABL == 11\n__attribute__((amdgpu_waves_per_eu(5, 5)))\n#elif QSMC_ABL == 13\n__attribute__((amdgpu_waves_per_eu(4, 4)))\n#else\n__attribute__((amdgpu_waves_per_eu(D >= 1 && D <= 2 ? 6 : 1, 8)))\n#endif")
s = rep(s, "#pragma unroll\n        for (int e = 0; e < 2; ++e) {\n            const int64_t o = 2 * P + e;\n            if (o >= o_begin && o < o_end) {\n                double p[DM];",
 "#if QSMC_ABL == 12\n#pragma nounroll\n#else\n#pragma unroll\n#endif\n        for (int e = 0; e < 2; ++e) {\n            const int64_t o = 2 * P + e;\n            if (o >= o_begin && o < o_end) {\n                double p[DM];")
s = rep(s, "        rng.uniforms(1, upos[0], upos[1]);",
 "#if QSMC_ABL == 4\n        upos[0] = (double)(P & 1023) * (1.0 / 1024.0); upos[1] = (double)((P * 7) & 1023) * (1.0 / 1024.0);\n#else\n        rng.uniforms(1, upos[0], upos[1]);\n#endif")
s = rep(s, "                nrm.normals(2, z[2 * k], z[2 * k + 1]);",
 "#if QSMC_ABL == 2\n                z[2 * k] = upos[0]; z[2 * k + 1] = upos[1];\n#else\n                nrm.normals(2, z[2 * k], z[2 * k + 1]);\n#endif")
s = rep(s, "            int j = use_guide ? guided_upper_bound(",
 "#if QSMC_ABL == 3\n            int j = (int)(upos[e] * len) + (u > 1e300);\n#else\n            int j = use_guide ? guided_upper_bound(")
s = rep(s, "            an.jl[e] = j > len - 1 ? len - 1 : j;", "#endif\n            an.jl[e] = j > len - 1 ? len - 1 : j;")
s = "#ifndef QSMC_ABL\n#define QSMC_ABL 0\n#endif\n" + s
# phase timing (QSMC_ABL == 30): per-workgroup cycle counts of load+scan / pair loop, read by qsmc_dbg_phase
s = rep(s, "struct BucketPlan {", "#if QSMC_ABL == 30\nextern \"C\" int qsmc_dbg_phase(double *out) { static unsigned long long h[8 * 8192]; if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_phase), sizeof(h)) != hipSuccess) return -1; for (int i = 0; i < 8; ++i) out[i] = 0; for (int b = 0; b < 8192; ++b) { if (h[2 * 8192 + b]) { out[0] += (double)h[b]; out[1] += (double)h[8192 + b]; out[2] += 1; out[3] += (double)h[3 * 8192 + b]; out[4] += (double)h[4 * 8192 + b]; out[5] += (double)h[5 * 8192 + b]; } } memset(h, 0, sizeof(h)); return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), h, sizeof(h)) == hipSuccess ? 0 : -1; }\n#endif\nstruct BucketPlan {")
s = rep(s, "constexpr int REDUCE_OUT_MAX = 192;", "constexpr int REDUCE_OUT_MAX = 192;\n#if QSMC_ABL == 30\n__device__ unsigned long long g_phase[8 * 8192];\n#endif")
s = rep(s, "    if ((int)blockIdx.x >= item_off[chunks]) return;\n    const int c = item_chunk[blockIdx.x];", "    if ((int)blockIdx.x >= item_off[chunks]) return;\n#if QSMC_ABL == 30\n    const unsigned long long ph0 = wall_clock64();\n#endif\n    const int c = item_chunk[blockIdx.x];")
s = rep(s, "    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;         // this item's output slots", "#if QSMC_ABL == 30\n    const unsigned long long ph1 = wall_clock64();\n#endif\n    const int64_t o_begin = slot0 + t0, o_end = slot0 + t1;         // this item's output slots")
s = rep(s, "    if (failed) atomicAdd(n_failed, failed);\n    __syncthreads();", "#if QSMC_ABL == 30\n    { const unsigned long long ph2 = wall_clock64(); if (threadIdx.x == 0 && blockIdx.x < 8192) { g_phase[blockIdx.x] = ph1 - ph0; g_phase[8192 + blockIdx.x] = ph2 - ph1; g_phase[2 * 8192 + blockIdx.x] = 1ull; } }\n#endif\n    if (failed) atomicAdd(n_failed, failed);\n    __syncthreads();")
s = rep(s, "    const int64_t i0 = c * SCAN_CHUNK + j0;\n    double v[SCAN_PER_LANE];", "    const int64_t i0 = c * SCAN_CHUNK + j0;\n#if QSMC_ABL == 30\n    const unsigned long long pa = wall_clock64();\n#endif\n    double v[SCAN_PER_LANE];")
s = rep(s, "        if (lane == QSMC_WAVE - 1) wave_tot[wave] = inc;\n    }\n    __syncthreads();", "        if (lane == QSMC_WAVE - 1) wave_tot[wave] = inc;\n    }\n#if QSMC_ABL == 30\n    const unsigned long long pb = wall_clock64();\n#endif\n    __syncthreads();\n#if QSMC_ABL == 30\n    const unsigned long long pc = wall_clock64();\n#endif")
s = rep(s, "        store(j, a, prev, j < len);\n        prev = a;\n    }\n}", "        store(j, a, prev, j < len);\n        prev = a;\n    }\n#if QSMC_ABL == 30\n    if (threadIdx.x == 0 && blockIdx.x < 8192) { const unsigned long long pd = wall_clock64(); g_phase[3 * 8192 + blockIdx.x] = pb - pa; g_phase[4 * 8192 + blockIdx.x] = pc - pb; g_phase[5 * 8192 + blockIdx.x] = pd - pc; }\n#endif\n}")
s = rep(s, "constexpr int UPD_UNROLL = 4;", "#if QSMC_ABL == 40\nconstexpr int UPD_UNROLL = 1;\n#elif QSMC_ABL == 41\nconstexpr int UPD_UNROLL = 2;\n#else\nconstexpr int UPD_UNROLL = 4;\n#endif")
s = rep(s, "    block_publish<UpdAcc<DMOM>::NS>(acc.s, acc.mn, ro);\n}\n\n// ---------------------------------------------------------------------------------------------\n// K data in ONE pass", "#if QSMC_ABL == 42\n    if (acc.s[0] == 1.2345e-300) block_publish<UpdAcc<DMOM>::NS>(acc.s, acc.mn, ro);\n#else\n    block_publish<UpdAcc<DMOM>::NS>(acc.s, acc.mn, ro);\n#endif\n}\n\n// ---------------------------------------------------------------------------------------------\n// K data in ONE pass")
s = rep(s, "template <int KIND, bool POW>\n__host__ __device__ __forceinline__ double model_lik(const double *p, const ExpArgs &e, int64_t o) {\n    const double L = Model<KIND>::lik(p, e, o);", "template <int KIND, bool POW>\n__host__ __device__ __forceinline__ double model_lik(const double *p, const ExpArgs &e, int64_t o) {\n#if QSMC_ABL == 43\n    const double L = p[0] * 0.5 + 0.25;\n#else\n    const double L = Model<KIND>::lik(p, e, o);\n#endif") if False else s
# variant 50: sampling loop only -- the chunk CDF and guide are synthetic (linear), no weights read, no scan
s = rep(s, "    chunk_scan_block(w, n_in, inv_norm, offsets, (int64_t)c, wave_tot,\n                     StoreLdsGuide{lcdf, lguide, lo_edge, gscale, use_guide, len, -1});",
 "#if QSMC_ABL == 50\n    for (int j = threadIdx.x; j < len; j += BT) lcdf[lds_skew(j)] = lo_edge + (hi_edge - lo_edge) * (double)(j + 1) / (double)len;\n    for (int k = threadIdx.x; k <= SGUIDE_BINS; k += BT) { long long g = ((long long)k * len + SGUIDE_BINS - 1) / SGUIDE_BINS - 1; lguide[k] = (unsigned short)(g < 0 ? 0 : (g > len ? len : g)); }\n#else\n    chunk_scan_block(w, n_in, inv_norm, offsets, (int64_t)c, wave_tot,\n                     StoreLdsGuide{lcdf, lguide, lo_edge, gscale, use_guide, len, -1});\n#endif")
# header variant: QSMC_ABL == 20 -> library log / sqrt / sincospi in Box-Muller
hdr = open(os.path.join(root, 'python-qinfer_amd/csrc/qsmc_device.h')).read()
hdr = rep(hdr, "        const double r = bm_sqrt(-2.0 * bm_log(1.0 - u0));  // 1 - u0 in [2^-53, 1]\n        double s, c;\n        bm_sincospi(2.0 * u1, s, c);",
 "#if QSMC_ABL == 20\n        const double r = sqrt(-2.0 * log(1.0 - u0));\n        double s, c;\n        sincospi(2.0 * u1, &s, &c);\n#else\n        const double r = bm_sqrt(-2.0 * bm_log(1.0 - u0));\n        double s, c;\n        bm_sincospi(2.0 * u1, s, c);\n#endif")
hdr = rep(hdr, "template <int KIND, bool POW>\n__host__ __device__ __forceinline__ double model_lik(const double *p, const ExpArgs &e, int64_t o) {\n    const double L = Model<KIND>::lik(p, e, o);", "template <int KIND, bool POW>\n__host__ __device__ __forceinline__ double model_lik(const double *p, const ExpArgs &e, int64_t o) {\n#if QSMC_ABL == 43\n    const double L = p[0] * 0.5 + 0.25;\n#else\n    const double L = Model<KIND>::lik(p, e, o);\n#endif")
tmph = os.path.join(root, 'python-qinfer_amd/csrc/_abl_dev.h')
open(tmph, 'w').write(hdr)
s = rep(s, '#include "qsmc_device.h"', '#include "_abl_dev.h"')
tmp = os.path.join(root, 'python-qinfer_amd/csrc/_abl_tmp.hip')
open(tmp, 'w').write(s)
os.makedirs(os.path.join(root, 'tools/abl_libs'), exist_ok=True)
procs = []
for n in sys.argv[1:]:
    out = os.path.join(root, 'tools/abl_libs/libqsmc_abl%s.so' % n)
    procs.append(subprocess.Popen(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
                                   '-fPIC', '-shared', '-DQSMC_ABL=%s' % n, '-I' + os.path.join(root, 'include'),
                                   tmp, '-o', out]))
rc = [p.wait() for p in procs]
os.remove(tmp)
os.remove(tmph)
print('built', sys.argv[1:], rc)
