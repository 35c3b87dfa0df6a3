# usage (GPU box): bash tools/sq_run.sh <config> [pass]  -- SQ counter passes over a short bench run -> gpurun_out/sq/<config>.json
# pass "lds": LDS conflict / stall counters only
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=${1:-cfg5}
out=gpurun_out/sq
mkdir -p $out
if [ "$2" = "lds" ]; then
timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL --kernel-trace -d $out -o c_$cfg -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $out/c.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE --kernel-trace -d $out -o d_$cfg -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $out/d.log 2>&1
python tools/sq_counters.py $out/${cfg}_lds.json $out/c_${cfg}_results.db $out/d_${cfg}_results.db
else
timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace -d $out -o a_$cfg -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $out/a.log 2>&1
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $out -o b_$cfg -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline > $out/b.log 2>&1
python tools/sq_counters.py $out/$cfg.json $out/a_${cfg}_results.db $out/b_${cfg}_results.db
fi
rm -f $out/*_results.db
