#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel trace of bench.py, prints the launches of the last training step.
#   tools/trace_step.sh NAME [extra bench args]   -> gpurun_out/prof_NAME/
name=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-profile --gen-steps 0 "$@" > gpurun_out/prof_$name.log 2>&1 || { tail -5 gpurun_out/prof_$name.log; exit 1; }
python - "$name" <<'PY'
import csv, glob, sys
f = glob.glob(f"gpurun_out/prof_{sys.argv[1]}/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
per = len(rows) // 7
for r in rows[-per:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if d > 30:
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "")
        print(f"{d:9.1f} us  {n[:60]}")
PY
