# usage (GPU box): bash tools/gemm_trace.sh <config>   -- per-launch GEMM shapes (MG_GEMM_TRACE) beside a one-step kernel timeline
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=${1:-cfg5}
out=gpurun_out/gemm_trace
mkdir -p $out
marker=k_prep_weights; if [ "$cfg" = "cfg2" ]; then marker=k_prep_lists; fi  # first kernel of a step (small batches: the merged launch)
MG_GEMM_TRACE=1 timeout -k 5 300 python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline 2> $out/shapes_$cfg.txt > /dev/null
grep -c "gemm\|dw" $out/shapes_$cfg.txt
timeout -k 5 300 rocprofv3 --kernel-trace -d $out -o prof_$cfg -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline > $out/log_$cfg.txt 2>&1
python tools/rocpd_timeline.py $out/prof_${cfg}_results.db $marker > $out/timeline_$cfg.txt 2>&1
rm -f $out/*_results.db
