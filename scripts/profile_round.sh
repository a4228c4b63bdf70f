# one GPU call: kernel-trace summaries of the bench (one stream / two streams) and the per-layer PMC traffic passes.
# usage (on the GPU box, from the repo root): bash scripts/profile_round.sh
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --serial-streams"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_serial -- $CMD > $R/gpurun_out/p_serial.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/p_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/p_default.log 2>&1
cd $R
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p_fetch -- python3 $R/scripts/traffic_layers.py 5 > $R/gpurun_out/p_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p_write -- python3 $R/scripts/traffic_layers.py 5 > $R/gpurun_out/p_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/p_l2 -- python3 $R/scripts/traffic_layers.py 5 > $R/gpurun_out/p_l2.log 2>&1 || true
python scripts/collect_traffic_layers.py gpurun_out/p_fetch gpurun_out/p_write gpurun_out/traffic_plan.json gpurun_out/traffic_per_layer.json gpurun_out/p_l2 > gpurun_out/traffic_layers.log 2>&1
cp $(ls gpurun_out/p_serial/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_serial.csv
cp $(ls gpurun_out/p_default/*/*kernel_stats.csv | head -1) gpurun_out/kernel_stats_default.csv
cp $(ls gpurun_out/p_default/*/*kernel_trace.csv | head -1) gpurun_out/kernel_trace_default.csv
python scripts/trace_step.py gpurun_out/kernel_trace_default.csv > gpurun_out/step_trace.txt 2>&1 || true
rm -rf gpurun_out/p_serial gpurun_out/p_default gpurun_out/p_fetch gpurun_out/p_write gpurun_out/p_l2
tail -1 gpurun_out/p_serial.log | cut -c1-200
tail -1 gpurun_out/p_default.log | cut -c1-200
cat gpurun_out/traffic_layers.log
