#!/bin/bash
for n in "$@"; do
  if [ "$n" = "d" ]; then unset QSMC_ABL_LIB; else export QSMC_ABL_LIB=/root/repo/tools/abl_libs/libqsmc_abl$n.so; fi
  echo "== variant $n"
  bash /root/repo/tools/pmc_run.sh tools/abl2.py k_bucket_sample SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES
done
