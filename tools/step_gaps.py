"""Dev tool: from a rocprofv3 --kernel-trace CSV of bench.py, the device time of the last training step: sum of the
kernel durations, span from the first kernel's start to the last one's end, and the idle gaps between launches.
    python tools/step_gaps.py gpurun_out/prof_NAME/*/*kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])
# steps are delimited by the Nadam kernel
idx = [i for i, k in enumerate(ks) if "nadam" in k[2]]
for a, b in zip(idx[-4:-1], idx[-3:]):
    step = ks[a + 1:b + 1]
    busy = sum(e - s for s, e, _ in step)
    span = step[-1][1] - step[0][0]
    gaps = sorted(((step[i + 1][0] - step[i][1]) for i in range(len(step) - 1)), reverse=True)
    print(f"step: {len(step)} launches, busy {busy / 1e6:.3f} ms, span {span / 1e6:.3f} ms, idle {(span - busy) / 1e6:.3f} ms; "
          f"largest gaps (us): {[round(g / 1e3, 1) for g in gaps[:6]]}, median {gaps[len(gaps) // 2] / 1e3:.1f}")
