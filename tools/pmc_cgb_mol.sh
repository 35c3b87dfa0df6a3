# usage (GPU box): bash tools/pmc_cgb_mol.sh [config]  -- FETCH_SIZE / WRITE_SIZE of the CG adjoint launches, default form vs the
# molecule-stationary form (MG_CGB_MOL=0), one counter per pass (--kernel-trace only), raw KiB -> MB per launch (FETCH x2: gfx950)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
cfg=${1:-cfg5}
for mol in off 0; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmccg; mkdir -p /tmp/pmccg
  if [ "$mol" = "off" ]; then env="MG_X=0"; else env="MG_CGB_MOL=$mol"; fi
  env $env timeout -k 5 300 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmccg -o f -- python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-epoch-overlap --no-build > /tmp/pmccg/log 2>&1
  python - <<PY
import sqlite3, glob
db = glob.glob('/tmp/pmccg/**/f_results.db', recursive=True)[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(counter_value) from pmc_events where counter_name='$ctr' group by name").fetchall()
for n, cnt, t in rows:
    if 'k_catbuild_bwd' in n:
        f = 2.0 if '$ctr' == 'FETCH_SIZE' else 1.0
        print('MG_CGB_MOL=$mol', '$ctr', n.split('(')[0][:30], cnt, 'launches', round(f * t * 1024 / 1e6 / cnt, 1), 'MB per launch')
PY
done
done
