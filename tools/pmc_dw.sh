# (the MG_DW_SHARE / MG_DW_BALANCE switches live in profiles/r04_experiments/dw_share_sets.patch: apply it first)
# usage (GPU box): bash tools/pmc_dw.sh <config>  -- FETCH_SIZE of the weight-gradient launches with share sets off / on
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=${1:-cfg5}
for sh in 0 1; do
  rm -rf /tmp/pmcdw; mkdir -p /tmp/pmcdw
  MG_DW_SHARE=$sh MG_DW_BALANCE=0 timeout -k 5 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmcdw -o f -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-epoch-overlap --no-build > /tmp/pmcdw/log 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmcdw/**/f_results.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name='FETCH_SIZE' group by name").fetchall()
tot = sum(r[2] for r in rows)
print('MG_DW_SHARE=$sh total fetch (raw KiB -> GB, all launches of the run)', round(tot * 1024 / 1e9, 2))
for n, cnt, t in sorted(rows, key=lambda r: -r[2])[:8]:
    print('   ', n.split('(')[0][:40], cnt, 'launches', round(t * 1024 / 1e6 / cnt, 1), 'MB raw per launch')
PY
done
