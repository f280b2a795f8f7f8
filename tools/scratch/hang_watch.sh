#!/bin/bash
# hang_watch.sh LIMIT_SECONDS OUTFILE pytest-args...   : runs pytest under rocgdb; if it is still running after LIMIT seconds the
# inferior is interrupted and host backtraces + GPU queue / dispatch / wave state are written to OUTFILE (and OUTFILE.hang is created).
limit=$1; out=$2; shift 2
rm -f "$out.hang"
/opt/rocm/bin/rocgdb -q -batch -ex "set pagination off" -ex "set confirm off" -ex "handle SIGINT stop print nopass" -ex "run" \
  -ex "echo \n==== HOST THREADS ====\n" -ex "thread apply all bt 14" -ex "echo \n==== AGENTS/QUEUES/DISPATCHES ====\n" -ex "info agents" -ex "info queues" -ex "info dispatches" \
  -ex "echo \n==== WAVES ====\n" -ex "info threads" -ex "kill" \
  --args python -X faulthandler -m pytest "$@" > "$out" 2>&1 &
gdbpid=$!
( sleep "$limit"; touch "$out.hang"; for c in $(pgrep -P "$gdbpid"); do kill -INT "$c"; done ) > /dev/null 2>&1 &
watcher=$!
wait "$gdbpid"
for c in $(pgrep -P "$watcher"); do kill "$c" 2>/dev/null; done
kill "$watcher" 2>/dev/null
if [ -e "$out.hang" ]; then echo "HANG (see $out)"; else grep -E "passed|failed" "$out" | head -2; fi
