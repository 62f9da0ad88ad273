for i in 1 2 3; do
for v in 0 1; do
  if [ $v = 1 ]; then export QSMC_NO_SPECULATIVE_PREFIX=1; else unset QSMC_NO_SPECULATIVE_PREFIX; fi
  python bench.py --steps 200 --warmup 20 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nospec' if os.environ.get('QSMC_NO_SPECULATIVE_PREFIX') else 'spec  ', d['ms_per_step'], d['config'].get('resamples'))"
done; done
