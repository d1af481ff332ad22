#!/bin/bash
# one trace per value of an env knob, grep one kernel: AB_VAR=NAME AB_VALUES="a b" KERNEL=regex
for v in ${AB_VALUES}; do
  echo "${AB_VAR}=$v"
  env "${AB_VAR}=$v" bash tools/gpu_trace.sh "trace_${AB_VAR}_$v" | grep -i -E "${KERNEL}"
done
