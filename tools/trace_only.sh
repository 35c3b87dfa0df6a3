# kernel trace of a short bench run -> gpurun_out/<tag>/{prof_results.db,kernel_stats.csv,timeline.txt}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-trace}; cfg=${2:-cfg2}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 5 200 rocprofv3 --kernel-trace -d $out -o prof -- python bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_prof.log 2>&1
python tools/rocpd_summary.py $out/prof_results.db $out/kernel_stats.csv 13 > /dev/null 2>&1
python tools/rocpd_timeline.py $out/prof_results.db > $out/timeline.txt 2>&1
tail -1 $out/bench_prof.log | cut -c1-200
