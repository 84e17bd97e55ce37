R=$PWD; cd /tmp; export TMPDIR=/tmp; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_gpt2_parity.py -m gpu -x -q -k "attention or attn or engine or block" > gpurun_out/t_attn.log 2>&1; tail -2 gpurun_out/t_attn.log
for v in old new old new; do
  cp ab/lib_$v.so drakegpt_amd/lib/libdrakegpt_hip.so
  rm -rf gpurun_out/prof_ab_$v
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ab_$v -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-extra > gpurun_out/prof_ab_$v.log 2>&1 || exit 1
  echo "== $v"; python tools/step_trace.py gpurun_out/prof_ab_$v/*/*_kernel_trace.csv | grep -E "step span|attn_"
  rm -rf gpurun_out/prof_ab_$v
done
cp ab/lib_new.so drakegpt_amd/lib/libdrakegpt_hip.so
