#!/bin/bash
# usage: bash tools/trace_extra.sh NAME CONFIG B PREC [ENV=VAL ...] -- kernel trace of tools/extra_bench.py, last step's kernel summary -> gpurun_out/NAME.txt
name=$1; cfg=$2; b=$3; prec=$4; shift 4
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $R/gpurun_out/prof_$name
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_$name -- python3 $R/tools/extra_bench.py $cfg $b $prec 3 2 > $R/gpurun_out/prof_$name.log 2>&1 || exit $?
f=$(ls $R/gpurun_out/prof_$name/*/*_kernel_trace.csv | head -1)
python3 $R/tools/step_trace.py $f > $R/gpurun_out/$name.txt
rm -rf $R/gpurun_out/prof_$name
cat $R/gpurun_out/$name.txt
