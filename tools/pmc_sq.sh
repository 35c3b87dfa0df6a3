cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/sq
mkdir -p $out
timeout -k 5 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $out -o sq -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/sq.log 2>&1
timeout -k 5 200 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $out -o sq2 -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/sq2.log 2>&1
python - <<'PY'
import sqlite3
for db in ('gpurun_out/sq/sq_results.db','gpurun_out/sq/sq2_results.db'):
    try:
        c=sqlite3.connect(db)
        rows=c.execute("select name, counter_name, count(*), avg(counter_value) from pmc_events group by name, counter_name").fetchall()
    except Exception as e:
        print(db,'ERR',e); continue
    d={}
    for n,cn,cnt,avg in rows:
        n=n.split('(')[0].replace('void ','')
        d.setdefault(n,{})[cn]=avg
    for k in ('k_catbuild_bwd','k_heads_bwd','k_heads_fwd','k_catbuild','k_gemm_dw<20>','k_gemm_rows<20, 4, 4>','k_gemm_cols<20>'):
        if k in d: print(k, {a:round(b) for a,b in d[k].items()})
PY
tail -3 $out/sq2.log | cut -c1-300
