# usage: ab_lib.sh <cfg> libA.so libB.so   -- same-box A/B of two builds (paths relative to the repo root)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/ab
mkdir -p $out
cfg=$1; shift
for rep in 1 2 3; do
for lib in "$@"; do
  MOLGYM_HIP_LIB=$GRAFT_REPO_ROOT/$lib BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-build > $out/b.json 2> $out/b.err
  tail -1 $out/b.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$lib', round(d['value']), round(d['ms_per_step'],4))" || tail -3 $out/b.err
done
done
