#!/usr/bin/env python3
"""Scan a gfx950 assembly listing for v_permlane16_swap_b32 instructions that read an MFMA result too soon.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -S --cuda-device-only drakegpt_amd/csrc/gemm.hip -o /tmp/gemm.s
    python tools/mfma_swap_scan.py /tmp/gemm.s [min_slots]

The GEMM epilogues exchange accumulator halves between lane rows through inline asm, and the compiler inserts no MFMA-result
wait states in front of inline asm (found in round 3 as a timing-dependent 3 % error in csrc/chain.hip's plain epilogue, where
nothing else stood between the last MFMAs and the first swap).  For every swap this counts issue slots (instructions, s_nop n
as n + 1) back to the MFMA that last wrote one of its registers and lists those closer than min_slots (default 12; an 8-pass
v_mfma_f32_16x16x32_bf16 needs 10)."""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
limit = int(sys.argv[2]) if len(sys.argv) > 2 else 12
worst, func, last_mfma, n_instr = {}, None, [], 0
for i, l in enumerate(lines):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        func, last_mfma, n_instr = m.group(1), [], 0
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    n_instr += 1
    mm = re.match(r"v_mfma\S+ v\[(\d+):(\d+)\]", t)
    if mm:
        last_mfma = (last_mfma + [(n_instr, int(mm.group(1)), int(mm.group(2)))])[-40:]
        continue
    if t.startswith("s_nop"):
        n_instr += int(t.split()[1])
        continue
    pm = re.match(r"v_permlane16_swap_b32 v(\d+), v(\d+)", t)
    if pm:
        for r in (int(pm.group(1)), int(pm.group(2))):
            for idx, lo, hi in reversed(last_mfma):
                if lo <= r <= hi:
                    if n_instr - idx < limit:
                        worst.setdefault(func, []).append((n_instr - idx, i + 1))
                    break
for f, v in worst.items():
    print(f[:90], sorted(v)[:5], len(v))
print("functions with an MFMA -> swap distance below", limit, "issue slots:", len(worst))
sys.exit(1 if worst else 0)
