# usage (GPU box): bash tools/ab_env_internal.sh <lib.so> "ENV=val ..." ...  -- SchNetAC bench under different environment switches
cd $GRAFT_REPO_ROOT
lib=$1; shift
for envs in "$@"; do
  env $envs MOLGYM_HIP_LIB=$PWD/$lib python bench.py --agent internal --steps 50 --warmup 10 --no-cpu-baseline --no-build 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$envs', round(d['value']), 'ms', round(d['ms_per_step'], 4))"
done
