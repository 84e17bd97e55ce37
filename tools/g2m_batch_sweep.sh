#!/bin/bash
# GPT-2-medium shape, bf16 vs fp8 at several batch sizes (one box)
for B in 8 16 32; do
  for prec in bf16 fp8; do
    echo "== gpt2_medium B=$B $prec"
    timeout -k 10 280 python tools/extra_bench.py gpt2_medium $B $prec 5 2 2>/dev/null | tail -1 || exit 1
  done
done
