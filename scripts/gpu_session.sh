# one GPU call = a list of steps; a step that times out or is killed stops the call (no further GPU work behind a hung kernel).
# usage: bash scripts/gpu_session.sh "<name>|<seconds>|<command>" ...    (output of each step: gpurun_out/<name>.log)
mkdir -p gpurun_out
for spec in "$@"; do
  name="${spec%%|*}"; rest="${spec#*|}"; secs="${rest%%|*}"; cmd="${rest#*|}"
  echo "== $name ($secs s): $cmd"
  timeout -k 10 "$secs" bash -c "$cmd" > "gpurun_out/$name.log" 2>&1
  rc=$?
  echo "== $name rc=$rc"; tail -n 6 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "== $name timed out / was killed: stopping the call"; exit $rc; fi
done
