# host-side helper (build container): retries a gpurun call while the pod has no free GPU slot (nothing is charged for those)
# usage: bash scripts/gpurun_retry.sh <timeout> '<command>'
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2" > /tmp/gpurun_last.out 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.out; then sleep 90; continue; fi
  cat /tmp/gpurun_last.out; exit $rc
done
cat /tmp/gpurun_last.out; exit 3
