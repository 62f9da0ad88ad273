cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/r4n; mkdir -p $out
QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --hip-trace --output-format csv -d /tmp/tl2 -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 60 --warmup 10 --no-other-configs --no-cpu-baseline > $out/c2.log 2>&1
kt=$(find /tmp/tl2 -name "*kernel_trace.csv"); ht=$(find /tmp/tl2 -name "*hip_api_trace.csv")
python $GRAFT_REPO_ROOT/tools/timeline.py $kt $ht "k_bucket_sample<1" 12 200 400 | grep -v "hipGetLastError\|CallConfiguration" > $out/c2_timeline.txt
head -70 $out/c2_timeline.txt
