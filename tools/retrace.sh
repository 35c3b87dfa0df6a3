# usage (GPU box): bash tools/retrace.sh <tag> <config>...  -- kernel trace only (stats with medians + timeline) of the given configs
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/$tag; mkdir -p $out
for config in "$@"; do
  marker=k_prep_weights; if [ "$config" = "cfg2" ]; then marker=k_prep_lists; fi
  timeout -k 5 300 rocprofv3 --kernel-trace -d $out -o prof_$config -- python bench.py --config $config --steps 20 --warmup 3 --no-cpu-baseline --no-epoch-overlap > $out/bench_prof_$config.log 2>&1
  python tools/rocpd_summary.py $out/prof_${config}_results.db $out/kernel_stats_$config.csv 43 > /dev/null && head -4 $out/kernel_stats_$config.csv
  python tools/rocpd_timeline.py $out/prof_${config}_results.db $marker > $out/timeline_$config.txt 2>&1
  rm -f $out/*_results.db
done
