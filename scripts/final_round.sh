# the measurements of the final code of a round, in two GPU calls (each within gpurun's limit):
#   bash scripts/final_round.sh a    bench line + rocprofv3 summaries (one stream / two streams, mangled names) + the bench-vs-rocprofv3 join + PMC passes
#   bash scripts/final_round.sh b    per-layer replay, other BASELINE configurations, data-parallel rows, sampler
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out
if [ "$1" = "a" ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 400 $O/bench_default.json
bash scripts/profile_round.sh > $O/profile_round.log 2>&1 || true
python scripts/compare_bench_rocprof.py $O/bench_default.json $O/kernel_stats_serial.csv > $O/bench_vs_rocprof.txt || true
cat $O/bench_vs_rocprof.txt
else
python scripts/engine_layers.py > $O/layers.txt 2>&1
python bench.py --size 256 --batch 16 --dtype f16 --no-cpu-baseline > $O/bench_config5.json 2>/dev/null
python bench.py --size 64 --batch 32 --no-cpu-baseline > $O/bench_config2.json 2>/dev/null
python scripts/bench_dp_overhead.py 40 > $O/dp_overhead.txt 2>&1 || true
python scripts/bench_sampler.py > $O/sampler.txt 2>&1 || true
tail -3 $O/layers.txt; tail -c 300 $O/bench_config5.json; tail -c 300 $O/bench_config2.json
fi
echo done
