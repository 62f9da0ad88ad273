#!/bin/bash
# round 5: the GPU suite, then same-box A/B of the list pass of a 2-qubit canonicalize (eigenvector-free form against
# QSMC_CANON_JACOBI=1) on the config-5 share, and a kernel trace of it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5b
mkdir -p $O
timeout 2000 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for rep in 1 2 3; do
  for j in 0 1; do
    if [ $j = 1 ]; then export QSMC_CANON_JACOBI=1; else unset QSMC_CANON_JACOBI; fi
    timeout 300 python bench.py --only config5_share_tomography --warmup 5 > $O/c5_jac${j}_$rep.json 2>$O/err.log
    python - <<PY
import json
d=json.load(open("$O/c5_jac${j}_$rep.json"))["config5_share_tomography"]
print("C5 jacobi=$j rep=$rep ms/step %.5f resamples %d canon_list %.1f us kick %.1f anc %.1f mom %.1f mean %s" % (d["ms_per_step"], d["resamples"], d["canonicalize"]["canon_list_us"], d["resample_kernel"]["kick_us"], d["resample_kernel"]["ancestors_us"], d["moments_kernel"]["avg_kernel_us"], d["posterior_mean_head"]))
PY
  done
done
unset QSMC_CANON_JACOBI
cd /tmp
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_c5 -o c5 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --only config5_share_tomography --warmup 5 > $GRAFT_REPO_ROOT/$O/prof_c5.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_c5 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/c5_kernel_stats.csv
find $O/prof_c5 -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/trace_gaps.py {} > $O/c5_trace_gaps.txt 2>&1
rm -rf $O/prof_c5
head -14 $O/c5_kernel_stats.csv | cut -c1-90,200-400
head -12 $O/c5_trace_gaps.txt
