# Host + device timeline of the d = 16 resample (config 5 share): rocprofv3 --hip-trace --kernel-trace, no PMC.
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r4c
mkdir -p $out
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl -- python $GRAFT_REPO_ROOT/bench.py --only config5_share_tomography --warmup 5 > $out/c5_under_rocprof.log 2>&1
find /tmp/tl -name "*kernel_trace.csv" -exec cp {} $out/c5_kernel_trace.csv \;
find /tmp/tl -name "*hip_api_trace.csv" -exec cp {} $out/c5_hip_api_trace.csv \;
python $GRAFT_REPO_ROOT/tools/timeline.py $out/c5_kernel_trace.csv $out/c5_hip_api_trace.csv k_moments_mfma 3 > $out/c5_timeline.txt 2>&1
head -c 6000 $out/c5_timeline.txt
ls -la $out
# keep the merge under the cap
gzip -f $out/c5_hip_api_trace.csv $out/c5_kernel_trace.csv
