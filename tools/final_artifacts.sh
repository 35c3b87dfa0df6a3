# usage (GPU box, through gpurun): bash tools/final_artifacts.sh <tag>
# full GPU test suite, contract bench line, kernel trace (+ per-kernel summary and one-step timeline) and the two
# PMC passes (FETCH_SIZE / WRITE_SIZE, each in its own run with --kernel-trace only) -> gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 5 500 python -m pytest tests/ -m gpu -q 2>&1 | grep -E "^E  |passed|failed|Error" | cut -c1-400 | tee $out/pytest.log
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out -o fetch -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/fetch.log 2>&1
timeout -k 5 150 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out -o write -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $out/write.log 2>&1
python tools/pmc_summary.py $out/fetch_results.db $out/write_results.db $out/pmc.json | head -8
cp $out/pmc.json profiles/pmc_latest.json   # bench.py reads the traffic of the dominant kernel from here
BENCH_WATCHDOG=250 timeout -k 5 300 python bench.py --steps 30 --warmup 5 > $out/bench.json 2> $out/bench.err
tail -1 $out/bench.json | cut -c1-1500
timeout -k 5 150 rocprofv3 --kernel-trace -d $out -o prof -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_prof.log 2>&1
python tools/rocpd_summary.py $out/prof_results.db $out/kernel_stats.csv 43 > /dev/null && head -16 $out/kernel_stats.csv && tail -1 $out/kernel_stats.csv
python tools/rocpd_timeline.py $out/prof_results.db k_prep_weights > $out/timeline.txt 2>&1
rm -f $out/fetch_results.db $out/write_results.db $out/prof_results.db
