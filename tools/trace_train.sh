# usage (GPU box): bash tools/trace_train.sh [rollout] [mini batch]  -- kernel timeline of ONE ppo.train epoch (cfg2)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
out=gpurun_out/gemm_trace
mkdir -p $out
timeout -k 5 300 rocprofv3 --kernel-trace -d $out -o prof_train -- python tools/train_bench.py cfg2 ${1:-140} ${2:-140} 7 > $out/log_train.txt 2>&1
python tools/rocpd_timeline.py $out/prof_train_results.db k_prep_lists > $out/timeline_train.txt 2>&1
rm -f $out/*_results.db
