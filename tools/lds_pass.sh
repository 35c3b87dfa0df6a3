# usage (GPU box): bash tools/lds_pass.sh <config> <tag> [lib.so]  -- ONE counter pass (the LDS one of tools/stall_run.sh) over a short
# bench run of a library build: LDS instructions, array cycles and bank-conflict cycles per kernel -> gpurun_out/lds/<tag>.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=$1; tag=$2; lib=$3
out=gpurun_out/lds
mkdir -p $out
if [ -n "$lib" ]; then export MOLGYM_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
timeout -k 5 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $out -o ${tag} -- python bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-epoch-overlap --no-build > $out/$tag.log 2>&1 || tail -3 $out/$tag.log
python tools/sq_counters.py $out/$tag.json $out/${tag}_results.db
python - $out/$tag.json > $out/$tag.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for name, c in sorted(d.items()):
    if 'catbuild' not in name and 'heads' not in name:
        continue
    g = lambda k: c.get(k, 0.0)
    ia, w = g('SQ_LDS_IDX_ACTIVE'), max(g('SQ_WAVES'), 1)
    print(f"{name[:60]:60s} LDS instr/wave {g('SQ_INSTS_LDS') / w:7.1f}  array cycles/instr {ia / max(g('SQ_INSTS_LDS'), 1):5.2f}  "
          f"array cycles/wave {ia / w:8.1f}  bank-conflict share {100 * g('SQ_LDS_BANK_CONFLICT') / max(ia, 1):5.1f} %  "
          f"wave cycles stalled on LDS issue {100 * g('SQ_WAIT_INST_LDS') / max(g('SQ_WAVE_CYCLES'), 1):5.1f} %")
PY
cat $out/$tag.txt
rm -f $out/*_results.db
