cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/scale
mkdir -p $out
for cfg in cfg3 cfg4 cfg5; do
  BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  tail -1 $out/bench_$cfg.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], d['value'], d['ms_per_step'], d['config']['step_tflops_ragged'], d['roofline']['span_ms_per_step'])" || tail -3 $out/bench_$cfg.err
done
# the DP code path with one rank (RCCL init, all-reduce of the flat gradient)
BENCH_WATCHDOG=200 timeout -k 5 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_dp1.json 2> $out/bench_dp1.err
tail -1 $out/bench_dp1.json | cut -c1-200; tail -2 $out/bench_dp1.err
nvidia-smi 2>/dev/null | head -1; rocm-smi --showmeminfo vram 2>/dev/null | head -6
