python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k 'wgrad' 2>&1 | tail -2
python -m pytest tests/test_step_gpu.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2; do
echo "== r03 order"; GCT2_WG_R03=1 python scripts/bench_phases.py 40 | grep "^step"
echo "== turns + aligned"; python scripts/bench_phases.py 40 | grep "^step"
done
echo "== layers, new"; python scripts/engine_layers.py --only U0,U1,U2,D1,D2,D3 | grep -i 'wgrad'
echo "== layers, r03"; GCT2_WG_R03=1 python scripts/engine_layers.py --only U0,U1,U2,D1,D2,D3 | grep -i 'wgrad'
echo "== phases: r03 stage order"; GCT2_WG_R03=1 python scripts/stamp_clock.py --seconds 0.3 --layers U0.wgrad,U1.wgrad,D1.wgrad | grep -v "^#"
echo "== phases: waves take turns + scalar address code"; python scripts/stamp_clock.py --seconds 0.3 --layers U0.wgrad,U1.wgrad,D1.wgrad | grep -v "^#"
