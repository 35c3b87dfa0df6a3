cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/quick
mkdir -p $out
timeout -k 5 300 python -m pytest tests/test_gpu_forward.py tests/test_gpu_backward.py tests/test_gpu_internal.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-400
for cfg in cfg2 cfg3 cfg5; do
  BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  tail -1 $out/bench_$cfg.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:5], round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['roofline']['span_ms_per_step'].items()})" || tail -3 $out/bench_$cfg.err
done
