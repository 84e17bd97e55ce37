#!/bin/bash
# usage: bash tools/trace_step.sh NAME [ENV=VAL ...]  -- rocprofv3 kernel trace of a short bench run, last step's kernel sequence -> gpurun_out/NAME.txt
name=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_$name
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extra > $R/gpurun_out/prof_$name.log 2>&1 || exit $?
f=$(ls $R/gpurun_out/prof_$name/*/*_kernel_trace.csv | head -1)
python3 $R/tools/step_trace.py $f --seq > $R/gpurun_out/$name.txt
rm -rf $R/gpurun_out/prof_$name
tail -25 $R/gpurun_out/$name.txt
