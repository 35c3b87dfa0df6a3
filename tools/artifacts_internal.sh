# usage (GPU box): bash tools/artifacts_internal.sh <tag> [steps]  -- BASELINE configs[0] (SchNet internal-coordinate agent):
# bench line, kernel stats and one-step timeline -> gpurun_out/<tag>/  (copy into profiles/<tag>_*_internal.*)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r03}
steps=${2:-20}
out=gpurun_out/$tag
mkdir -p $out
BENCH_WATCHDOG=300 timeout -k 5 400 python bench.py --agent internal --steps 50 --warmup 10 > $out/bench_internal.json 2> $out/bench_internal.err
tail -1 $out/bench_internal.json | cut -c1-1500
timeout -k 5 300 rocprofv3 --kernel-trace -d $out -o prof_internal -- python bench.py --agent internal --steps $steps --warmup 3 --no-cpu-baseline --no-build > $out/bench_prof_internal.log 2>&1
python tools/rocpd_summary.py $out/prof_internal_results.db $out/kernel_stats_internal.csv $((steps + 3)) > /dev/null && head -24 $out/kernel_stats_internal.csv && tail -1 $out/kernel_stats_internal.csv
python tools/rocpd_timeline.py $out/prof_internal_results.db k_int_fill_lists > $out/timeline_internal.txt 2>&1
rm -f $out/*_results.db
