cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/prof1
BENCH_WATCHDOG=280 timeout 300 python bench.py --steps 30 --warmup 5 > gpurun_out/prof1/b1.log 2>&1
tail -3 gpurun_out/prof1/b1.log | cut -c1-3000
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof1 -o r1 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/prof1/bench_prof.log 2>&1
find gpurun_out/prof1 -name "*stats*" | head
f=$(find gpurun_out/prof1 -name "*kernel_stats*" | head -1)
head -40 $f
