# usage (GPU box): bash tools/ab_libs3.sh <config> <steps> <reps> libA.so libB.so ...  -- large configs, interleaved repetitions, one line per run
cd $GRAFT_REPO_ROOT
cfg=$1; steps=$2; reps=$3; shift; shift; shift
for rep in $(seq $reps); do for lib in "$@"; do
MOLGYM_HIP_LIB=$PWD/$lib python bench.py --config $cfg --steps $steps --warmup 3 --no-cpu-baseline --no-build --no-epoch-overlap 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib', '$cfg', round(d['value']), round(d['ms_per_step'], 4), 'median', round(d['config']['median_ms_per_step'], 4))"
done; done
