#!/bin/bash
# same-box A/B of two builds of the library: OLD_LIB=<path to the other libqsmc_hip.so> (QSMC_LIB_PATH selects it)
for i in 1 2 3; do
  echo -n "new "; python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
  echo -n "old "; QSMC_LIB_PATH=${OLD_LIB:-/root/repo/tools/_ab/libqsmc_old.so} python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
