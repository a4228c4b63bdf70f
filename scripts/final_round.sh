set -e
cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
tail -c 600 gpurun_out/bench_default.json
bash scripts/profile_round.sh > gpurun_out/profile_round.log 2>&1 || true
python scripts/stamp_clock.py --seconds 1.0 --layers D3.fwd,D4.fwd,D5.fwd,U5.fwd,U4.fwd,U3.fwd,U3.dgrad,U4.dgrad,U5.dgrad,D5.dgrad,D4.dgrad,D3.dgrad,U3.wgrad,U4.wgrad,U5.wgrad,D5.wgrad,D4.wgrad,D3.wgrad,U0.wgrad,U1.wgrad,U2.wgrad,D1.wgrad,D2.wgrad,U0.fwd,U1.fwd,U2.fwd,D1.dgrad,D2.dgrad,U0.dgrad,U1.dgrad,U2.dgrad,D1.fwd,D2.fwd > gpurun_out/kernel_clock.txt 2>&1
python scripts/engine_layers.py > gpurun_out/layers.txt 2>&1
python scripts/engine_layers.py --zeros > gpurun_out/layers_zero_data.txt 2>&1
( echo "# per-stage phases of wgrad256q_kernel's steady-state K loop (diagnostic build: make phases; cycles per 32-row stage, mean over the waves of a group)"
  echo "## r05 order: waves take turns, the next stage's fragments read under the multiplies (tuning 0; 'reads' is part of '32 MFMA' here)"
  python scripts/stamp_clock.py --phases --seconds 0.5 --layers U0.wgrad,U1.wgrad,U2.wgrad,D1.wgrad,D2.wgrad,D3.wgrad | grep -v amdgpu.ids
  echo "## r04 order: waves take turns, reads up front, scalar stage position (tuning bits 16-23 = 5)"
  python scripts/stamp_clock.py --phases --tuning 0x50000 --seconds 0.5 --layers U0.wgrad,U1.wgrad,U2.wgrad,D1.wgrad,D2.wgrad,D3.wgrad | grep -v "amdgpu.ids\|^#"
  echo "## r03 order (tuning bits 16-23 = 4)"
  python scripts/stamp_clock.py --phases --tuning 0x40000 --seconds 0.5 --layers U0.wgrad,U1.wgrad,U2.wgrad,D1.wgrad,D2.wgrad,D3.wgrad | grep -v "amdgpu.ids\|^#" ) > gpurun_out/wgrad_stage_phases.txt 2>&1 || true
tests/hw_probe/probe_power > gpurun_out/probe_power.txt 2>&1 || true
python bench.py --size 256 --batch 16 --dtype f16 --no-cpu-baseline > gpurun_out/bench_config5.json 2>/dev/null
python bench.py --size 64 --batch 32 --no-cpu-baseline > gpurun_out/bench_config2.json 2>/dev/null
python scripts/bench_dp_overhead.py 30 > gpurun_out/dp_overhead.txt 2>&1 || true
python scripts/bench_sampler.py > gpurun_out/sampler.txt 2>&1 || true
echo done
