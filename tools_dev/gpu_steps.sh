#!/bin/bash
# Run GPU steps one after another on the gpurun box; stop at the first step that crashed or
# timed out (exit code > 1), continue after ordinary test failures (exit code 1).
# usage: gpu_steps.sh "<cmd1>" "<cmd2>" ...
mkdir -p gpurun_out
for cmd in "$@"; do
  echo "=== $(date +%T) $cmd"
  bash -c "$cmd"
  rc=$?
  echo "=== rc=$rc"
  if [ $rc -gt 1 ]; then echo "stopping: step crashed or timed out"; exit $rc; fi
done
exit 0
