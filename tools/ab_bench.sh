#!/bin/bash
# same-box A/B of library builds: tools/ab_bench.sh libX.so libY.so ...   (paths relative to the repo; "-" = in-tree)
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset QSMC_LIB_PATH; else export QSMC_LIB_PATH=/root/repo/$lib; fi
  python /root/repo/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs | tail -1 | python -c "
import json,sys; d=json.load(sys.stdin); c=d['kernel_census']['kernels']
print('$lib', 'p-u/s %.3e' % d['value'], 'ms/step %.4f' % d['ms_per_step'], 'sample %.1f us' % d['resample_kernel']['avg_kernel_us'], 'counts %.1f' % c['counts']['avg_us'], 'upd %.1f' % c['update']['avg_us'], 'mean %.6f' % d['posterior_mean'], d['config']['resamples_in_timed_region'])"
done
