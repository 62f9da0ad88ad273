#!/bin/bash
# round 6: per-kernel A/B of the builds of the library (in-tree: -fgpu-rdc + 8 LTO partitions; gpurun_ab/libqsmc_rdc_o3.so: the
# same with --lto-O3; gpurun_ab/libqsmc_nordc.so: one plain device compile) -- LTO must not cost any kernel anything
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6f; mkdir -p $O
for cfg in config5_share_tomography config4_share_rb; do
  for lib in rdc rdc_o3 nordc; do
    if [ $lib = rdc ]; then unset QSMC_LIB_PATH; else export QSMC_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_ab/libqsmc_$lib.so; fi
    bash tools/kstats.sh --only $cfg --warmup 5 > $O/${cfg}_${lib}.txt 2>&1
    echo "== $cfg $lib"; grep -E "anc16|kick16|sample_ordered|bank|update_fused|update_tomo|canon|moments|chunk_scan|redraw" $O/${cfg}_${lib}.txt | cut -c1-100
  done
done
unset QSMC_LIB_PATH
