# usage: ab_env.sh <cfg> VAR val1 val2 ...   (bench with VAR=val for each val, twice)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/ab
mkdir -p $out
cfg=$1; var=$2; shift 2
for rep in 1 2; do
for v in "$@"; do
  env $var=$v BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline > $out/b.json 2> $out/b.err
  tail -1 $out/b.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$var=$v', round(d['value']), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['roofline']['span_ms_per_step'].items()})" || tail -3 $out/b.err
done
done
