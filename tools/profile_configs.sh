#!/bin/bash
# Run ON THE GPU BOX (via gpurun): per-config rocprofv3 evidence for profiles/<tag>_*.
#   <tag>_bench_line.json              the default bench line (headline + other_configs + beyond-L3 + cpu_baseline)
#   <tag>_c2_kernel_stats.csv          rocprofv3 --kernel-trace --stats of the headline command (200 steps)
#   <tag>_{c3,c4,c5}_kernel_stats.csv  the same for `bench.py --only <config>`
#   <tag>_{c2,c3,c4,c5}_pmc_{fetch,write}.csv   separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel-trace only)
#   <tag>_{c2,c3,c4,c5}_trace_gaps.txt  idle gaps between kernels of the stats run (tools/trace_gaps.py)
#   <tag>_paths_kernel_stats.csv       `bench.py --only other_paths` (batch_update windows, bayes_risk)
#   <tag>_step_host_time.txt           host time per datum: qsmc_step vs the round-2 Python path
# usage: tools/profile_configs.sh <tag> [pmc]
tag=${1:-r2_x}
want_pmc=${2:-}
out=/root/repo/gpurun_out/profiles_new
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
python /root/repo/bench.py --steps 200 --warmup 20 2>$out/${tag}_bench.err | tail -1 > $out/${tag}_bench_line.json
prof() {   # name, args...
  name=$1; shift
  rm -rf $out/ks
  # (QSMC_BENCH_NO_EVENTS: no HIP events on the launches -- an event-carrying launch drains the queue around itself and
  #  would show up as gaps that the un-instrumented run does not have)
  QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -- python /root/repo/bench.py "$@" > $out/${tag}_${name}_under_rocprof.log 2>&1
  cp $(ls $out/ks/*/*kernel_stats.csv | tail -1) $out/${tag}_${name}_kernel_stats.csv
  # idle gaps between kernels from the same run's kernel trace (tools/trace_gaps.py)
  kt=$(ls $out/ks/*/*kernel_trace.csv 2>/dev/null | tail -1)
  if [ -n "$kt" ]; then
    { echo "# tools/trace_gaps.py over rocprofv3 --kernel-trace of: bench.py $@ (the profiler inflates every gap by a few us)";
      python /root/repo/tools/trace_gaps.py $kt; } > $out/${tag}_${name}_trace_gaps.txt 2>&1
  fi
  if [ -n "$want_pmc" ]; then
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf $out/pmc
      rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc -- python /root/repo/bench.py "$@" > /dev/null 2>&1
      cp $(ls $out/pmc/*/*counter_collection.csv | tail -1) $out/${tag}_${name}_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).csv
    done
  fi
  rm -rf $out/ks $out/pmc
}
prof c2 --steps 200 --warmup 20 --no-cpu-baseline --no-other-configs
prof c3 --only config3_binomial_precession --warmup 5
prof c4 --only config4_share_rb --warmup 5
prof c5 --only config5_share_tomography --warmup 5
prof paths --only other_paths
{ python /root/repo/tools/step_host_time.py; QSMC_NO_STEP=1 python /root/repo/tools/step_host_time.py; } > $out/${tag}_step_host_time.txt 2>&1
if [ -n "$want_pmc" ]; then
  python /root/repo/tools/make_traffic_json.py $out $tag > $out/hbm_traffic.json
fi
head -c 1500 $out/${tag}_bench_line.json; echo
for c in c2 c4 c5; do echo "== gaps $c"; head -12 $out/${tag}_${c}_trace_gaps.txt | cut -c1-150; done
cat $out/${tag}_step_host_time.txt
for c in c2 c3 c4 c5 paths; do echo "== $c"; head -8 $out/${tag}_${c}_kernel_stats.csv | cut -c1-150; done
