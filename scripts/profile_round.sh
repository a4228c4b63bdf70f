# one GPU call: kernel-trace summaries of the bench (one stream / two streams), the PMC passes over the ENGINE'S OWN launches
# (traffic, L2 hit rate, MFMA utilisation: separate --pmc runs of scripts/engine_layers.py --pmc K) and the in-kernel clock of the K
# loops (diagnostic library).  usage (on the GPU box, from the repo root): bash scripts/profile_round.sh [tag]
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
K=5
EL="python3 $R/scripts/engine_layers.py --pmc $K"
if [ "$1" != "pmc-only" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --mangled-kernels --output-format csv -d $O/p_serial -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --serial-streams > $O/p_serial.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --mangled-kernels --output-format csv -d $O/p_default -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/p_default.log 2>&1
fi
cd $R
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- $EL > $O/p_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- $EL > $O/p_write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p_l2 -- $EL > $O/p_l2.log 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p_mfma -- $EL > $O/p_mfma.log 2>&1
python scripts/collect_engine_pmc.py traffic $O/engine_plan.json $O/traffic_per_layer.json $O/p_fetch $O/p_write $O/p_l2 > $O/traffic_layers.log 2>&1
python scripts/collect_engine_pmc.py mfma $O/engine_plan.json $O/mfma_util.txt $O/p_mfma > $O/mfma_util.log 2>&1
if [ "$1" != "pmc-only" ]; then
cp $(ls $O/p_serial/*/*kernel_stats.csv | head -1) $O/kernel_stats_serial.csv
cp $(ls $O/p_default/*/*kernel_stats.csv | head -1) $O/kernel_stats_default.csv
cp $(ls $O/p_default/*/*kernel_trace.csv | head -1) $O/kernel_trace_default.csv
python scripts/trace_step.py $O/kernel_trace_default.csv 15 > $O/step_trace.txt      # (step 15 = the middle of the timed region of --warmup 5 --steps 20; later steps carry the event legs' records) 2>&1 || true
tail -1 $O/p_serial.log | cut -c1-200
tail -1 $O/p_default.log | cut -c1-200
fi
rm -rf $O/p_serial $O/p_default $O/p_fetch $O/p_write $O/p_l2 $O/p_mfma
cat $O/traffic_layers.log | tail -45
cat $O/mfma_util.log | tail -40
