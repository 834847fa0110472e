#!/bin/bash
# Stability check of the two stream-ordering modes: short bench runs, exit codes only (crashes print the tail of stderr).
for i in 1 2 3 4; do
  for m in 1 0; do
    CREAM_FORK_ON_KERNEL=$m timeout 100 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-host-leg > /tmp/s.json 2> /tmp/s.err
    rc=$?
    echo "fork_on_kernel=$m run $i rc $rc"
    if [ $rc -ne 0 ]; then grep -n "Fatal\|fault\|Fault\|Error\|error" /tmp/s.err | head -5; head -c 1500 /tmp/s.err; fi
  done
done
