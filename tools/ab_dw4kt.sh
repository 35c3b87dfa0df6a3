cd $GRAFT_REPO_ROOT
for cfg in cfg5 cfg4 cfg3; do
for kt in 1 2 3; do
  MG_DW4_KT=$kt MOLGYM_HIP_LIB=$PWD/ab_build/lib_v7.so python bench.py --config $cfg --steps 20 --warmup 3 --no-cpu-baseline --no-build --no-epoch-overlap 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('kt$kt', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 3), {k: round(v * 1e3, 1) for k, v in r['span_ms_per_step'].items() if 'dw' in k})"
done
done
