cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in ab_build/lib_nq0.so ab_build/lib_nq4.so ab_build/lib_nq8.so; do
MOLGYM_HIP_LIB=$PWD/$lib python bench.py --agent internal --steps 60 --warmup 10 --no-cpu-baseline --no-build 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$lib internal', round(d['value']), round(d['ms_per_step'], 4))"
done; done
bash tools/ab_libs2.sh cfg2 2 ab_build/lib_nq0.so ab_build/lib_nq4.so ab_build/lib_nq8.so
