#!/usr/bin/env python3
"""Summarise the last training step of a rocprofv3 --kernel-trace CSV: kernel sequence with durations.
    python tools/step_trace.py gpurun_out/prof_x/<host>/<pid>_kernel_trace.csv [--seq]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adamw')]
seq = rows[idx[-2] + 1:idx[-1] + 1]
agg = collections.OrderedDict()
for i, r in enumerate(seq):
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'][:58]
    if '--seq' in sys.argv:
        print(i, n, f"{d:.1f}")
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d
span = (int(seq[-1]['End_Timestamp']) - int(seq[0]['Start_Timestamp'])) / 1e3
print(f"step span {span:.0f} us, {len(seq)} kernels, busy {sum(v[1] for v in agg.values()):.0f} us")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:58s} x{c:3d}  {t:8.1f} us  avg {t / c:7.1f}")
