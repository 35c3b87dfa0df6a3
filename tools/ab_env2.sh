# usage (GPU box): bash tools/ab_env2.sh <lib.so> <config> <reps> "ENV=val ENV2=val" ...   -- same library, different environment
# switches, the timed headline leg only (no live roofline spans: graph launches as the driver times them), interleaved repetitions
cd $GRAFT_REPO_ROOT
lib=$1; cfg=$2; reps=$3; shift; shift; shift
for rep in $(seq $reps); do
for envs in "$@"; do
  env $envs MOLGYM_HIP_LIB=$PWD/$lib python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-build --no-epoch-overlap 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$envs', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'median', round(d['config']['median_ms_per_step'], 4), 'launches', d['config']['kernel_launches_per_step'])"
done
done
