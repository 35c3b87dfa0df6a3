cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/ab
mkdir -p $out
for rep in 1 2; do
for side in 0 1; do
  MG_NO_SIDE_STREAM=$side BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --config ${1:-cfg2} --steps 30 --warmup 5 --no-cpu-baseline > $out/b.json 2> $out/b.err
  tail -1 $out/b.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('no_side=$side', round(d['value']), round(d['ms_per_step'],3), 'host', round(d['config']['host_enqueue_ms_per_step'],3))" || tail -3 $out/b.err
done
done
