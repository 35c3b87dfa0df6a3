# usage (GPU box): bash tools/quick_trace.sh <lib.so> <config>  -- kernel trace of a short bench run with that library: per-kernel stats head
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
lib=$1; config=${2:-cfg2}; out=gpurun_out/qt; mkdir -p $out
MOLGYM_HIP_LIB=$PWD/$lib timeout -k 5 300 rocprofv3 --kernel-trace -d $out -o qt -- python bench.py --config $config --steps 20 --warmup 3 --no-cpu-baseline --no-epoch-overlap --no-build > $out/log.txt 2>&1
python tools/rocpd_summary.py $out/qt_results.db $out/stats.csv 43 > /dev/null && head -${3:-24} $out/stats.csv
rm -f $out/*_results.db
