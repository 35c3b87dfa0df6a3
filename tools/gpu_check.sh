# usage: bash tools/gpu_check.sh <tag>   (run on the GPU box through gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-x}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 5 400 python -m pytest tests/ -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-600 | tee $out/pytest.log
BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-1200
timeout -k 5 150 rocprofv3 --kernel-trace -d $out -o prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_prof.log 2>&1
ls $out
python tools/rocpd_summary.py $out/prof_results.db $out/kernel_stats.csv 43 && head -25 $out/kernel_stats.csv && tail -1 $out/kernel_stats.csv
