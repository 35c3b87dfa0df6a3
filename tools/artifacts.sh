# usage (GPU box, through gpurun): bash tools/artifacts.sh <tag> [config] [steps]
# contract bench line, kernel trace (+ per-kernel summary and one-step timeline) and the two PMC passes
# (FETCH_SIZE / WRITE_SIZE, each in its own run with --kernel-trace only) of ONE config -> gpurun_out/<tag>/
# (copy what should be judged into profiles/: <tag>_kernel_stats_<config>.csv, <tag>_timeline_<config>.txt,
#  pmc_<config>.json -- bench.py reads the dominant kernel's HBM traffic from profiles/pmc_<config>.json)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-r02}
config=${2:-cfg2}
steps=${3:-20}
warm=3
out=gpurun_out/$tag
mkdir -p $out
marker=k_prep_weights; if [ "$config" = "cfg2" ]; then marker="void k_level0_fwd"; fi  # first kernel of a step (small batches: the merged launch)
timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o fetch_$config -- python bench.py --config $config --steps 5 --warmup 2 --no-cpu-baseline --no-epoch-overlap > $out/fetch_$config.log 2>&1
timeout -k 5 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out -o write_$config -- python bench.py --config $config --steps 5 --warmup 2 --no-cpu-baseline --no-epoch-overlap > $out/write_$config.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out -o sq_$config -- python bench.py --config $config --steps 5 --warmup 2 --no-cpu-baseline --no-epoch-overlap > $out/sq_$config.log 2>&1
PMC_STEPS=27 python tools/pmc_summary.py $out/fetch_${config}_results.db $out/write_${config}_results.db $out/pmc_$config.json $out/sq_${config}_results.db | head -8
cp $out/pmc_$config.json profiles/pmc_$config.json
# (the CPU-baseline leg is the contract's cfg2 line only: one oracle pass at the larger configs takes minutes)
extra=""; if [ "$config" != "cfg2" ]; then extra="--no-cpu-baseline"; fi
BENCH_WATCHDOG=400 timeout -k 5 500 python bench.py --config $config --steps 50 --warmup 10 $extra > $out/bench_$config.json 2> $out/bench_$config.err
tail -1 $out/bench_$config.json | cut -c1-2500
timeout -k 5 300 rocprofv3 --kernel-trace -d $out -o prof_$config -- python bench.py --config $config --steps $steps --warmup $warm --no-cpu-baseline --no-epoch-overlap > $out/bench_prof_$config.log 2>&1
# launches in the trace: warm-up + timed steps + the 20 iterations of the live roofline leg
python tools/rocpd_summary.py $out/prof_${config}_results.db $out/kernel_stats_$config.csv $((steps + warm + 20)) > /dev/null && head -14 $out/kernel_stats_$config.csv && tail -1 $out/kernel_stats_$config.csv
# a TIMED step (the trace ends with the 20 event-bracketed steps of the live roofline leg)
python tools/rocpd_timeline.py $out/prof_${config}_results.db "$marker" 22 > $out/timeline_$config.txt 2>&1
rm -f $out/*_results.db
