# usage (GPU box): bash tools/ab_r6.sh <config> <reps> libA.so libB.so ...   -- same-box A/B of library builds: step time and the
# live per-family spans (bench.py roofline leg); paths relative to the repo root
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=$1; reps=$2; shift; shift
for rep in $(seq $reps); do
for lib in "$@"; do
  MOLGYM_HIP_LIB=$GRAFT_REPO_ROOT/$lib BENCH_WATCHDOG=300 timeout -k 5 400 python bench.py --config $cfg --steps 40 --warmup 10 --no-cpu-baseline --no-epoch-overlap --no-build 2>/tmp/ab_err.log | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$lib', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 4), 'median', round(d['config']['median_ms_per_step'], 4), {k: round(v * 1e3, 1) for k, v in r['span_ms_per_step'].items()})" || tail -3 /tmp/ab_err.log
done
done
