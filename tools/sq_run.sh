# usage (GPU box): bash tools/sq_run.sh <config>  -- two SQ counter passes over a short bench run -> gpurun_out/sq/<config>.json
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=${1:-cfg5}
out=gpurun_out/sq
mkdir -p $out
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d $out -o a_$cfg -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $out/a.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out -o b_$cfg -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $out/b.log 2>&1
python tools/sq_counters.py $out/$cfg.json $out/a_${cfg}_results.db $out/b_${cfg}_results.db
rm -f $out/*_results.db
