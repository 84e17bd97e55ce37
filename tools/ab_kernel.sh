# usage: bash tools/ab_kernel.sh VAR a b PATTERN  -- per-kernel average (us) of kernels matching PATTERN under VAR=a / VAR=b, one box
R=$PWD; cd /tmp; export TMPDIR=/tmp; cd $R
for v in $2 $3; do
  rm -rf gpurun_out/prof_ab_$v
  env $1=$v rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab_$v -- python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/prof_ab_$v.log 2>&1
  echo "== $1=$v"; python tools/step_trace.py gpurun_out/prof_ab_$v/*/*_kernel_trace.csv | grep -E "step span|$4"
done
