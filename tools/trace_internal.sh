# usage (GPU box): bash tools/trace_internal.sh  -- one-step kernel timeline of the SchNet internal-coordinate agent
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/gemm_trace
mkdir -p $out
timeout -k 5 300 rocprofv3 --kernel-trace -d $out -o prof_int -- python bench.py --agent internal --steps 3 --warmup 1 --no-cpu-baseline > $out/log_int.txt 2>&1
python tools/rocpd_timeline.py $out/prof_int_results.db ${1:-k_int} > $out/timeline_int.txt 2>&1
python - <<'PY'
import sqlite3
c = sqlite3.connect('gpurun_out/gemm_trace/prof_int_results.db')
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
print([t for t in tabs if 'kernel' in t][:6])
PY
rm -f $out/*_results.db
