# usage (GPU box): bash tools/stall_run.sh <config> [steps]  -- stall attribution of the CG kernels: five SQ / cache counter
# passes (8 SQ slots each, --kernel-trace only) over a short bench run -> gpurun_out/stall/<config>.json + <config>.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=${1:-cfg2}
steps=${2:-3}
sel="--config $cfg"; if [ "$cfg" = "internal" ]; then sel="--agent internal"; fi  # (SchNetAC: BASELINE configs[0])
out=gpurun_out/stall
mkdir -p $out
run() {  # name counters...
  name=$1; shift
  timeout -k 5 400 rocprofv3 --pmc "$@" --kernel-trace -d $out -o ${name}_$cfg -- python bench.py $sel --steps $steps --warmup 1 --no-cpu-baseline --no-epoch-overlap --no-build > $out/$name.log 2>&1 || tail -3 $out/$name.log
}
run p1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE
run p2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR
run p3 SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run p4 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INST_LEVEL_SMEM
run p5 TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum
python tools/sq_counters.py $out/$cfg.json $out/p1_${cfg}_results.db $out/p2_${cfg}_results.db $out/p3_${cfg}_results.db $out/p4_${cfg}_results.db $out/p5_${cfg}_results.db
python tools/stall_table.py $out/$cfg.json $STALL_KERNELS > $out/$cfg.txt
cat $out/$cfg.txt
rm -f $out/*_results.db
