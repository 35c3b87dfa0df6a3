cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${1:-tint}
mkdir -p $out
timeout -k 5 200 rocprofv3 --kernel-trace -d $out -o prof -- python bench.py --agent internal --steps 10 --warmup 5 --no-cpu-baseline > $out/log.txt 2>&1
python tools/rocpd_summary.py $out/prof_results.db $out/kernel_stats.csv 15 > /dev/null 2>&1
head -30 $out/kernel_stats.csv; tail -1 $out/kernel_stats.csv
