"""one extra_configs line of bench.py on its own:  python tools/extra_bench.py gpt2_medium 8 fp8 [steps] [warmup]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

if __name__ == "__main__":
    cfg, B, prec = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
    warm = int(sys.argv[5]) if len(sys.argv) > 5 else 2
    r = bench.run_extra(cfg, B, None, steps, warm, torch.device("cuda:0"), prec)
    print(json.dumps({k: r[k] for k in ("value", "ms_per_step", "mfma_peak_frac_whole_step", "final_loss", "workload")}))
