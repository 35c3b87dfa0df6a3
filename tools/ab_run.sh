# usage (GPU box): bash tools/ab_run.sh <config> name1 name2 ...   -- bench the library variants ab_build/lib_<name>.so
cd $GRAFT_REPO_ROOT
cfg=$1; shift
for name in "$@"; do
  MOLGYM_HIP_LIB=$PWD/ab_build/lib_$name.so python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-build 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('$name', '$cfg', round(d['value']), 'ms', round(d['ms_per_step'], 3), {k: round(v * 1e3, 1) for k, v in r['span_ms_per_step'].items()})"
done
