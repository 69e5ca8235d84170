"""Instruction-class histogram of the largest loop of one kernel in a hipcc -S listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only csrc/dj_lstm.hip -o /tmp/dj_lstm.s
    python tools/isa_loop_hist.py /tmp/dj_lstm.s 'lstm_bwd_kernelIDF16bLi128ELb0ELi1'
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith(("ds_read", "ds_load")):
        return "lds_read"
    if op.startswith(("ds_write", "ds_store")):
        return "lds_write"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("global_load", "buffer_load", "flat_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store")):
        return "vmem_store"
    if op.startswith(("global_atomic", "buffer_atomic")):
        return "vmem_atomic"
    if op.startswith(("scratch_",)):
        return "scratch"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_accvgpr"):
        return "acc_mov"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().split(";")[0].strip().endswith(":"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(body):
        m = re.match(r"\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)", l) or re.match(r"\s+s_branch\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    def hist(rng):
        h = collections.Counter()
        ops = collections.Counter()
        for l in body[rng[0]:rng[1] + 1]:
            m = re.match(r"^\s+([a-z_0-9]+)", l)
            if m and not l.strip().startswith((".", ";")):
                h[classify(m.group(1))] += 1
                ops[m.group(1)] += 1
        return h, ops
    for name, rng in (("kernel", (0, len(body) - 1)), ("largest loop", best)):
        if rng is None:
            continue
        h, ops = hist(rng)
        print(f"== {name}: lines {rng[0]}..{rng[1]}, {sum(h.values())} instructions")
        for k, v in sorted(h.items(), key=lambda kv: -kv[1]):
            print(f"   {k:12s} {v}")
        if name != "kernel" and "-v" in sys.argv:
            for k, v in ops.most_common(60):
                print(f"      {k:28s} {v}")


if __name__ == "__main__":
    main()
