#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes over bench.py into profiles/traffic.json (HBM bytes per launch).

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing
    python tools/collect_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write

Units and gfx950 correction as guides/MI355X_MICROARCH.md prescribes (section HBM): the counters are in
KiB-ish units of 1024 B (hbm_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024) and FETCH_SIZE tallies the 128-B
requests of a wide coalesced stream at 64 B, i.e. reads exactly half the bytes: it is doubled here.
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        a = acc[r["Kernel_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, {k: v[1] for k, v in acc.items()}


def short(name):
    """kernel symbol as bench.py spells it: gemm_nt_ws_kernel<bf16,false,6>, gemm_tn_grouped256_kernel, ..."""
    m = re.match(r"_Z\d+(\w+?)I((?:DF16b|f|Lb[01]E|Li\d+E)+)Ev", name)      # hipcc leaves __bf16 template args mangled
    if m:
        out = []
        for tok in re.findall(r"DF16b|f|Lb[01]E|Li\d+E", m.group(2)):
            if tok == "DF16b": out.append("bf16")
            elif tok == "f": out.append("float")
            elif tok.startswith("Lb"): out.append("true" if tok[2] == "1" else "false")
            else: out.append(tok[2:-1])
        return f"{m.group(1)}<{','.join(out)}>"
    name = re.sub(r"\(.*", "", name)
    name = name.replace("void ", "").replace(", ", ",")
    # rocprofv3's demangler garbles __bf16 / true template arguments: "<bool _Accum,bool,E,6,4>" is <bf16,true,6,4>
    name = name.replace("bool _Accum", "bf16").replace(",bool,E", ",true")
    return name.strip()


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    fetch, n = per_kernel(fd, "FETCH_SIZE")
    write, _ = per_kernel(wd, "WRITE_SIZE")
    out = {}
    for k in fetch:
        fb, wb = fetch[k] * 1024.0, write.get(k, 0.0) * 1024.0
        out[short(k)] = {"launches": n[k], "fetch_size_raw_bytes": fb, "write_size_bytes": wb,
                         "hbm_bytes_per_launch": 2.0 * fb + wb}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "profiles", "traffic.json"), "w") as f:
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py; FETCH_SIZE doubled (gfx950)",
                   "kernels": out}, f, indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:12]:
        print(f"{k[:60]:60s} launches={v['launches']:5d} hbm MB/launch={v['hbm_bytes_per_launch'] / 1e6:8.2f}")


if __name__ == "__main__":
    main()
