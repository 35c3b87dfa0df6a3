# kernel-trace summaries + bench lines of the larger BASELINE configs -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-large}
out=gpurun_out/$tag
mkdir -p $out
for cfg in cfg3 cfg4 cfg5; do
  BENCH_WATCHDOG=250 timeout -k 5 300 python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_$cfg.json 2> $out/bench_$cfg.err
  tail -1 $out/bench_$cfg.json | cut -c1-200
  timeout -k 5 200 rocprofv3 --kernel-trace -d $out/$cfg -o prof -- python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $out/prof_$cfg.log 2>&1
  python tools/rocpd_summary.py $out/$cfg/prof_results.db $out/kernel_stats_$cfg.csv 23 > /dev/null 2>&1
  rm -rf $out/$cfg
done
