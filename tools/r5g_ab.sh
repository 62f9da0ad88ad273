#!/bin/bash
# round 5: A/B on the config-5 share: in-tree build against gpurun_ab/lib_occ4.so (k_tomo_canon_list_fast held to 128 VGPRs:
# four waves per SIMD, 30 registers spilled); the publish kernels with the fence in the writing lane only
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5g
mkdir -p $O
for rep in 1 2 3; do
  for which in new occ4; do
    if [ $which = occ4 ]; then export QSMC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/lib_occ4.so; else unset QSMC_LIB_PATH; fi
    timeout 300 python bench.py --only config5_share_tomography --warmup 5 > $O/c5_${which}_$rep.json 2>$O/err.log
    python - <<PY
import json
d=json.load(open("$O/c5_${which}_$rep.json"))["config5_share_tomography"]
print("C5 $which rep=$rep ms/step %.5f resamples %d canon_list %.1f us kick %.1f anc %.1f mom %.1f" % (d["ms_per_step"], d["resamples"], d["canonicalize"]["canon_list_us"], d["resample_kernel"]["kick_us"], d["resample_kernel"]["ancestors_us"], d["moments_kernel"]["avg_kernel_us"]))
PY
  done
done
unset QSMC_LIB_PATH
cd /tmp
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats -d /tmp/pc5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --only config5_share_tomography --warmup 5 > /dev/null 2>&1
grep -h "sum_partials\|canon_list\|kick16" /tmp/pc5/*/*kernel_stats.csv | cut -c1-60,200-330
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats -d /tmp/pp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --only other_paths > $GRAFT_REPO_ROOT/$O/paths.json 2>/dev/null
grep -h "sum_columns\|hyp_sums\|publish" /tmp/pp/*/*kernel_stats.csv | cut -c1-60,200-330
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rank_ordered or design or g7 or tomo" 2>&1 | tail -3
