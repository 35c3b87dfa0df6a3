# usage (GPU box): bash tools/ab_env.sh <lib.so> <config> "ENV=val ENV2=val" ...   -- same library, different environment switches
cd $GRAFT_REPO_ROOT
lib=$1; cfg=$2; shift; shift
for envs in "$@"; do
  env $envs MOLGYM_HIP_LIB=$PWD/$lib python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-build --no-epoch-overlap 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$envs', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 3), {k: round(v * 1e3, 1) for k, v in r['span_ms_per_step'].items()})"
done
