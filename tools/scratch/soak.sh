#!/bin/bash
# soak.sh N LIMIT_SECONDS OUTDIR -- pytest args...   : N plain runs of the GPU suite; a run that exceeds LIMIT is examined
# (faulthandler's Python stacks are in its log; GPU use and the threads' kernel wait channels are appended) and then killed.
n=$1; limit=$2; out=$3; shift 3
mkdir -p "$out"
for i in $(seq 1 "$n"); do
  log="$out/run_$i.log"
  python -X faulthandler -m pytest "$@" -o faulthandler_timeout=$((limit - 30)) > "$log" 2>&1 &
  pid=$!
  t=0
  while kill -0 "$pid" 2>/dev/null && [ "$t" -lt "$limit" ]; do sleep 2; t=$((t + 2)); done
  if kill -0 "$pid" 2>/dev/null; then
    { echo "==== STALLED after $t s ===="; /opt/rocm/bin/rocm-smi --showuse --showmemuse 2>&1 | tail -12;
      for task in /proc/$pid/task/*; do echo "$(basename $task) $(cat $task/comm 2>/dev/null) wchan=$(cat $task/wchan 2>/dev/null) state=$(grep State $task/status 2>/dev/null | tr -s '\t ' ' ')"; done; } >> "$log"
    kill "$pid"; sleep 3; kill -9 "$pid" 2>/dev/null
    echo "iter $i: STALLED (see $log)"
    break
  fi
  wait "$pid"; rc=$?
  echo "iter $i rc=$rc $(grep -E 'passed|failed' "$log" | tail -1)"
  [ "$rc" -eq 0 ] && rm -f "$log"
done
