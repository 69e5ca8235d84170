#!/bin/bash
# Dev tool (GPU box): SQ counters of the recurrent sweeps inside a bench step (one pass per counter group, PMC with
# --kernel-trace only), per launch, summed over the chip:   tools/pmc_sweeps.sh  ->  gpurun_out/pmc_sweeps.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
args="--steps 2 --warmup 1 --cpu-sample 0 --no-profile --gen-steps 0 --no-fp32 --scaled-steps 0"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/pmc_sw_$i -- python bench.py $args > gpurun_out/pmc_sw_$i.log 2>&1 || { tail -5 gpurun_out/pmc_sw_$i.log; exit 1; }
done
python - <<'PY' | tee gpurun_out/pmc_sweeps.txt
import csv, glob, collections
names = {"lstm_fwd_cluster_kernelILb0ELi16": "fwd cluster (time L1)", "lstm_bwd256_kernel": "BPTT H=256 (split gate math)", "lstm_bwd_kernelIDF16bLi256": "BPTT H=256 (plain)", "lstm_fwd_fused_kernelIDF16bLi128ELb0ELb1": "fwd note L1", "lstm_bwd_kernelIDF16bLi128ELb0ELi1": "BPTT note L1", "lstm_wgrad_bf16": "wgrad"}
for d in sorted(glob.glob("gpurun_out/pmc_sw_*/")):
    fs = glob.glob(d + "*/*counter_collection.csv")
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(max(fs))):
        for k, nm in names.items():
            if k in r["Kernel_Name"]:
                acc[nm][r["Counter_Name"]] += float(r["Counter_Value"]); disp[nm].add(r["Dispatch_Id"])
    for nm in acc:
        for c, v in acc[nm].items():
            print(f"{nm:24s} {c:32s} {v/len(disp[nm]):18.0f} per launch ({len(disp[nm])} launches)")
PY
find gpurun_out -name "*kernel_trace.csv" -path "*pmc_sw_*" -delete
