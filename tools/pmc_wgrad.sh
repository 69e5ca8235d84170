#!/bin/bash
# Dev tool (GPU box): SQ / LDS counters of the isolated weight-gradient launch (tools/wgrad_bench.py), one pass per group.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_wg_$i -- python tools/wgrad_bench.py > gpurun_out/pmc_wg_$i.log 2>&1 || { tail -5 gpurun_out/pmc_wg_$i.log; exit 1; }
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_wg_*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(max(fs))):
        if "lstm_wgrad" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"{k:36s} {sum(v)/len(v):16.0f}  per launch ({len(v)} launches)")
PY
