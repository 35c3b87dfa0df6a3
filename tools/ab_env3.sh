# usage (GPU box): bash tools/ab_env3.sh <lib.so> <config> <steps> "ENV=val ENV2=val" ...   -- large configs: one run per switch set
# with the live span legs (family times per step)
cd $GRAFT_REPO_ROOT
lib=$1; cfg=$2; steps=$3; shift; shift; shift
for envs in "$@"; do
  env $envs MOLGYM_HIP_LIB=$PWD/$lib python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-build --no-epoch-overlap 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$envs', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 3), {k: round(v * 1e3) for k, v in r['span_ms_per_step'].items()})"
done
