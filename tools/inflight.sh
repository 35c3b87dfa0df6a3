cd $GRAFT_REPO_ROOT
for cfg in "140 1" "70 2" "47 3" "35 4" "70 1" "47 1"; do
  set -- $cfg
  BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --batch $1 --inflight $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch',d['config']['global_batch'],'inflight',d['config']['minibatches_in_flight'], round(d['value']), round(d['ms_per_step'],3))"
done
