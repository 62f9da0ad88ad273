#!/bin/bash
# round 6, last session: rocprofv3 evidence for the full-size single-GPU entries (C4 at 1e8, C5 at 1e7) -- kernel stats +
# separate --pmc FETCH_SIZE / WRITE_SIZE passes (kernel-trace only), as tools/profile_configs.sh does for the shares
tag=${T:-r6_e}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
cd /tmp; export TMPDIR=/tmp
prof() {
  name=$1; shift
  rm -rf $out/ks
  QSMC_BENCH_NO_EVENTS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks -- python /root/repo/bench.py "$@" > $out/${tag}_${name}_under_rocprof.log 2>&1
  cp $(ls $out/ks/*/*kernel_stats.csv | tail -1) $out/${tag}_${name}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/pmc
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $out/pmc -- python /root/repo/bench.py "$@" > /dev/null 2>&1
    cp $(ls $out/pmc/*/*counter_collection.csv | tail -1) $out/${tag}_${name}_pmc_$(echo $c | tr A-Z a-z | sed s/_size//).csv
  done
  rm -rf $out/ks $out/pmc
}
prof c4full --only config4_full_rb_1gpu --warmup 5
prof c5full --only config5_full_tomography_1gpu --warmup 5
for c in c4full c5full; do echo "== $c"; head -8 $out/${tag}_${c}_kernel_stats.csv | cut -c1-160; ls -la $out/${tag}_${c}_pmc_*.csv; done
