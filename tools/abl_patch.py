"""Build ablation variants of libqsmc_hip.so (tools/abl_libs/libqsmc_abl<N>.so) from a patched temp copy
of the kernel source.  Measurement tooling only; nothing here ships."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = open(os.path.join(root, 'python-qinfer_amd/csrc/qsmc_kernels.hip')).read()
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
s = rep(s, "__global__ __launch_bounds__(BT) void k_bucket_sample(",
 "#if QSMC_ABL == 6\n__attribute__((amdgpu_waves_per_eu(6, 6)))\n#endif\n__global__ __launch_bounds__(BT) void k_bucket_sample(")
s = "#ifndef QSMC_ABL\n#define QSMC_ABL 0\n#endif\n" + s
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
print('built', sys.argv[1:], rc)
