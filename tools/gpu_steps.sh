#!/bin/bash
# Run a list of GPU steps on the box, each under its own time limit, logging to gpurun_out/<tag>_<name>.log.
# A step that is killed by its limit (or by a signal) ends the session: nothing else touches the GPU after a hang.
#   tools/gpu_steps.sh TAG "name1|limit_s|command ..." "name2|limit_s|command ..."
TAG=$1; shift
mkdir -p gpurun_out
SUM=gpurun_out/${TAG}_summary.txt
: > $SUM
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; lim=${rest%%|*}; cmd=${rest#*|}
  echo "=== $name (limit ${lim}s): $cmd" | tee -a $SUM
  t0=$(date +%s)
  timeout -k 10 $lim bash -c "$cmd" > gpurun_out/${TAG}_${name}.log 2>&1
  rc=$?
  echo "    rc=$rc  $(( $(date +%s) - t0 ))s" | tee -a $SUM
  tail -n 3 gpurun_out/${TAG}_${name}.log | cut -c1-300 | sed 's/^/    | /' | tee -a $SUM
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "    step killed: stopping the session" | tee -a $SUM; exit 1; fi
done
exit 0
