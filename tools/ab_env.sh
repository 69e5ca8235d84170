#!/bin/bash
# Dev tool: A/B of the in-tree library under environment switches on the GPU box (bench.py without CPU baseline,
# fp32 leg, scaled record and generation).   tools/ab_env.sh base DEEPJ_BWD_PAIR=0 ...  ("base" = no switch)
for v in "$@"; do
  name=$(echo "$v" | tr '=' '_')
  if [ "$v" = base ]; then
    timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --gen-steps 0 --no-fp32 --scaled-steps 0 > gpurun_out/ab_$name.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab_$name.log; exit 1; }
  else
    env "$v" timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --gen-steps 0 --no-fp32 --scaled-steps 0 > gpurun_out/ab_$name.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab_$name.log; exit 1; }
  fi
  python - "$name" <<'PY'
import json, sys
v = sys.argv[1]
line = [l for l in open(f"gpurun_out/ab_{v}.log") if l.startswith("{")][-1]
d = json.loads(line); k = d["kernel_ms_per_step"]
print(v, "ms/step", d["ms_per_step"], "loss", d["final_loss"], {n: round(x, 2) for n, x in k.items() if n.startswith("lstm") or n.startswith("gemm")})
PY
done
