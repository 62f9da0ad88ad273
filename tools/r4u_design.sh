cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4u
for cfg in default CHAIN1; do
  case $cfg in
    default) envs="";;
    CHAIN1) envs="QSMC_HYP_CHAIN1=1";;
  esac
  echo "== $cfg"
  env $envs timeout 300 python3 tools/design_bench.py 1e7 gpurun_out/r4u/$cfg.npy 2>&1 | grep -v amdgpu.ids
done
timeout 900 python3 -m pytest tests -m gpu -x -q -k "bayes_risk or design or hyp or full_size or sharded or shard" 2>&1 | tail -3
timeout 600 python3 bench.py --only other_paths 2>/dev/null | python3 tools/other_paths_print.py
