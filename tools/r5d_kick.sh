#!/bin/bash
# round 5: A/B of k_bucket_kick16 with wave-level tile fences (this build) against workgroup barriers (gpurun_ab/lib_r5c.so),
# config-5 share, same box; then the failing / new tests only
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r5d
mkdir -p $O
for rep in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export QSMC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/lib_r5c.so; else unset QSMC_LIB_PATH; fi
    timeout 300 python bench.py --only config5_share_tomography --warmup 5 > $O/c5_${which}_$rep.json 2>$O/err.log
    python - <<PY
import json
d=json.load(open("$O/c5_${which}_$rep.json"))["config5_share_tomography"]
print("C5 $which rep=$rep ms/step %.5f resamples %d canon_list %.1f us kick %.1f anc %.1f mom %.1f mean %s" % (d["ms_per_step"], d["resamples"], d["canonicalize"]["canon_list_us"], d["resample_kernel"]["kick_us"], d["resample_kernel"]["ancestors_us"], d["moments_kernel"]["avg_kernel_us"], d["posterior_mean_head"]))
PY
  done
done
unset QSMC_LIB_PATH
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "window_kernels or tomography or tomo or step_path or d16 or design or full_size_other" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
