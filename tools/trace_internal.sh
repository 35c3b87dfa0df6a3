cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/${1:-tint}
mkdir -p $out
timeout -k 5 200 rocprofv3 --kernel-trace -d $out -o prof -- python tools/internal_bench.py 140 10 > $out/log.txt 2>&1
python tools/rocpd_summary.py $out/prof_results.db $out/kernel_stats.csv 15 > /dev/null 2>&1
head -30 $out/kernel_stats.csv; tail -1 $out/kernel_stats.csv
