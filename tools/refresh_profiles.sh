#!/bin/bash
# Run ON THE GPU BOX (via gpurun): regenerates everything under profiles/ into gpurun_out/profiles_new/<tag>_*.
#   bench line, rocprofv3 --kernel-trace --stats of the same command, and separate --pmc FETCH_SIZE / WRITE_SIZE passes.
tag=${1:-r1_x}
out=/root/repo/gpurun_out/profiles_new
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
python /root/repo/bench.py --steps 200 --warmup 20 2>/dev/null | tail -1 > $out/${tag}_bench_line.json
rm -rf $out/ks; rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -- python /root/repo/bench.py --steps 200 --warmup 20 > $out/${tag}_bench_under_rocprof.log 2>&1
cp $(ls $out/ks/*/*kernel_stats.csv | tail -1) $out/${tag}_kernel_stats_200steps.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc_$c -- python /root/repo/bench.py --steps 40 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
  cp $(ls $out/pmc_$c/*/*counter_collection.csv | tail -1) $out/${tag}_pmc_$(echo $c | tr A-Z a-z).csv
done
python /root/repo/tools/make_traffic_json.py $out/${tag}_pmc_fetch_size.csv $out/${tag}_pmc_write_size.csv > $out/hbm_traffic.json
rm -rf $out/ks $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE
tail -1 $out/${tag}_bench_under_rocprof.log | cut -c1-200
head -c 600 $out/hbm_traffic.json
