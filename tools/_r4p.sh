#!/bin/bash
set -o pipefail
python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r4p_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -4 gpurun_out/r4p_tests.log
[ $rc = 0 ] || exit 1
for v in counted tagged counted tagged; do
  if [ $v = counted ]; then export DEEPJ_TAGGED_EXCHANGE=0; else unset DEEPJ_TAGGED_EXCHANGE; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --gen-steps 0 --no-fp32 --scaled-steps 0 > gpurun_out/r4p_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/r4p_$v.log; exit 1; }
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
d = json.loads([l for l in open(f"gpurun_out/r4p_{v}.log") if l.startswith("{")][-1]); k = d["kernel_ms_per_step"]
print(v, "ms/step", d["ms_per_step"], "loss", d["final_loss"], "faults", d.get("cluster_faults"), {n: round(x, 2) for n, x in k.items() if n.startswith("lstm") or n.startswith("gemm")})
PY
done
