# usage: ab_env2.sh <cfg> VAR v1 v2 ... : prev lib once per round + current lib with VAR=v
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/ab
mkdir -p $out
cfg=$1; var=$2; shift 2
run() {
  BENCH_WATCHDOG=200 timeout -k 5 240 python bench.py --config $cfg --steps 40 --warmup 5 --no-cpu-baseline --no-build > $out/b.json 2> $out/b.err
  tail -1 $out/b.json | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$1', round(d['value']), round(d['ms_per_step'],4))" || tail -3 $out/b.err
}
for rep in 1 2 3; do
  MOLGYM_HIP_LIB=$GRAFT_REPO_ROOT/molgym_amd/libmg_prev.so run prev
  for v in "$@"; do export $var=$v; MOLGYM_HIP_LIB=$GRAFT_REPO_ROOT/molgym_amd/libmolgym_hip.so run "$var=$v"; unset $var; done
done
