# kernel-trace summary of the one-stream step (isolated launches) with MANGLED kernel names, joined with the bench line of the same box:
# usage (GPU box, repo root): bash scripts/rocprof_serial.sh <tag>   ->  gpurun_out/<tag>_kernel_stats.csv, <tag>_bench.json, <tag>_bench_vs_rocprof.txt
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=${1:-r06}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --mangled-kernels --output-format csv -d $O/p_serial -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-events --serial-streams > $O/${T}_serial.log 2>&1
cp $(ls $O/p_serial/*/*kernel_stats.csv | head -1) $O/${T}_kernel_stats.csv
rm -rf $O/p_serial
cd $R
python bench.py --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err
python scripts/compare_bench_rocprof.py $O/${T}_bench.json $O/${T}_kernel_stats.csv > $O/${T}_bench_vs_rocprof.txt
cat $O/${T}_bench_vs_rocprof.txt
