#!/bin/bash
# in-tree build of libqsmc_hip.so (same flags as __graft_entry__.build_library); prints errors and the resource usage of
# kernels whose name matches $1 (optional)
R=/root/repo
cd $R/python-qinfer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared qsmc_kernels.hip \
  -o $R/python-qinfer_amd/qinfer_amd/lib/libqsmc_hip.so -Rpass-analysis=kernel-resource-usage 2>/tmp/qsmc_build.txt
rc=$?
grep -E "error" -A4 /tmp/qsmc_build.txt | head -30
if [ -n "$1" ]; then
  grep -A10 "Function Name: .*$1" /tmp/qsmc_build.txt | grep -E "Function Name|VGPRs:|Occupancy|Spill|LDS|Scratch" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | paste - - - - - - - | cut -c1-260
fi
echo "build rc=$rc"
