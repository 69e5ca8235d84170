"""Summarise the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/rNN_x

writes <prefix>_pmc_hbm_per_kernel.csv and profiles/pmc_traffic.json (bench.py reads the latter).
Units per MI355X_MICROARCH.md: the counters are in KiB-like units of 1 KB; on gfx950 FETCH_SIZE tallies
128-byte requests at 64 bytes, so read bytes are doubled; WRITE_SIZE is taken as is.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)   # newest run
    acc = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def category(name):
    n = name
    if "lstm_bwd256_kernel" in n:
        return "lstm_bwd_time"
    if "lstm_bwd_kernel" in n:
        return "lstm_bwd_time" if "Li256" in n else "lstm_bwd_note"
    if "lstm_fwd_cluster" in n:
        return "lstm_fwd_time"
    if "lstm_fwd" in n:
        return "lstm_fwd_time" if "Li256" in n else "lstm_fwd_note"
    if "lstm_wgrad" in n:
        return "gemm_dw"
    if "gemm_nt_bf16" in n:
        return "gemm_nt_avg"
    return None


def main():
    fd, wd, prefix = sys.argv[1:4]      # e.g. gpurun_out/pmc_NAME/fetch gpurun_out/pmc_NAME/write profiles/rNN_x
    fe, wr = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    rows, cat = [], defaultdict(lambda: [0.0, 0])
    for k in sorted(fe, key=lambda k: -sum(fe[k])):
        n = len(fe[k])
        f_kb, w_kb = sum(fe[k]) / n, (sum(wr[k]) / len(wr[k]) if k in wr else 0.0)
        rd, wt = 2.0 * f_kb * 1024, w_kb * 1024
        rows.append((k, n, f_kb, w_kb, rd / 1e9, wt / 1e9))
        c = category(k)
        if c:
            cat[c][0] += (rd + wt) * n
            cat[c][1] += n
    with open(prefix + "_pmc_hbm_per_kernel.csv", "w") as f:
        f.write("kernel,launches,FETCH_SIZE_KB_avg,WRITE_SIZE_KB_avg,hbm_read_GB_corrected_x2,hbm_write_GB\n")
        for k, n, a, b, c, d in rows:
            if c + d > 0.005:
                f.write(f'"{k}",{n},{a:.1f},{b:.1f},{c:.3f},{d:.3f}\n')
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --steps 2 --warmup 1`; "
                     "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B); bytes per launch"}
    for c, (tot, n) in cat.items():
        out[c] = tot / n
    json.dump(out, open(os.path.join(os.path.dirname(prefix) or ".", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
