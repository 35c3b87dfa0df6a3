# (the MG_DW_SHARE / MG_DW_BALANCE switches live in profiles/r04_experiments/dw_share_sets.patch: apply it first)
# usage (GPU box): bash tools/ab_dw_share.sh <config> [steps]   -- weight-gradient share sets / balanced chunks on and off
cd $GRAFT_REPO_ROOT
cfg=${1:-cfg5}; steps=${2:-6}
for envs in "MG_DW_SHARE=0 MG_DW_BALANCE=0" "MG_DW_SHARE=1 MG_DW_BALANCE=0" "MG_DW_SHARE=0 MG_DW_BALANCE=1" "MG_DW_SHARE=1 MG_DW_BALANCE=1" "MG_DW_SHARE=0 MG_DW_BALANCE=0" "MG_DW_SHARE=1 MG_DW_BALANCE=1"; do
  env $envs python bench.py --config $cfg --steps $steps --warmup 2 --no-cpu-baseline --no-build --no-epoch-overlap 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$envs', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 3), {k: round(v * 1e3, 1) for k, v in r['span_ms_per_step'].items() if 'dw' in k or 'gemm' in k})"
done
