# usage (GPU box): bash tools/ab_libs2.sh <config> <reps> libA.so libB.so ...  -- the timed headline leg only (as ab_env2.sh), one line per library, interleaved
cd $GRAFT_REPO_ROOT
cfg=$1; reps=$2; shift; shift
for rep in $(seq $reps); do
for lib in "$@"; do
  MOLGYM_HIP_LIB=$PWD/$lib python bench.py --config $cfg --steps 60 --warmup 10 --no-cpu-baseline --no-build --no-epoch-overlap 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$lib', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'median', round(d['config']['median_ms_per_step'], 4), 'launches', d['config']['kernel_launches_per_step'])"
done
done
