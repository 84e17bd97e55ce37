# attention kernels at the GPT-2 shapes, equal-cost workgroup order (DG_ATTN_BALANCE=1) against heavy-first beyond one residency (2)
for cfg in "gpt2_medium 8" "gpt2_medium 16" "gpt2_small 8" "gpt2_small 16"; do set -- $cfg
  for m in 1 2; do echo "== $1 B=$2 DG_ATTN_BALANCE=$m"; DG_ATTN_BALANCE=$m timeout -k 10 200 python tools/kbench.py attn --cfg $1 --batch $2 2>/dev/null | grep "attention p=0.2" || exit 1; done
done
