cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/pmc
mkdir -p $out
BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-2500
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o fetch -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out -o write -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1
ls -la $out
python - <<'PY'
import sqlite3
c=sqlite3.connect('gpurun_out/pmc/fetch_results.db')
for v in ('pmc_events','counters_collection'):
    try:
        cur=c.execute(f'select * from {v} limit 2'); print(v,[d[0] for d in cur.description]); print(cur.fetchall())
    except Exception as e: print(v,'ERR',e)
PY
