#!/bin/bash
# dg_block_chain_bwd check, every mode; stops at the first timeout / kill (never start another GPU step after one)
mkdir -p gpurun_out
for mode in 2 1 0; do
  timeout -k 10 120 python tools/chain_bwd_check.py --mode $mode "$@" > gpurun_out/cbw_$mode.log 2>&1
  rc=$?
  echo "mode $mode rc $rc"; tail -25 gpurun_out/cbw_$mode.log
  if [ $rc -ge 124 ]; then exit $rc; fi
done
