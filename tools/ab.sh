#!/bin/bash
# Dev tool: time bench.py (no CPU baseline / generation) for several library variants on the GPU box.
#   tools/ab.sh base p1s1 ...  -> gpurun_out/ab_<name>.log, summary on stdout ("base" = the in-tree library)
for v in "$@"; do
  if [ "$v" = base ]; then unset DEEPJ_LIB; else export DEEPJ_LIB=$PWD/music-generator_amd/lib/libdeepj_hip.$v.so; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --gen-steps 0 --no-fp32 --scaled-steps 0 > gpurun_out/ab_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/ab_$v.log; exit 1; }
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
line = [l for l in open(f"gpurun_out/ab_{v}.log") if l.startswith("{")][-1]
d = json.loads(line); k = d["kernel_ms_per_step"]
print(v, "ms/step", d["ms_per_step"], "loss", d["final_loss"], {n: round(x, 2) for n, x in k.items() if n.startswith("lstm") or n.startswith("gemm")})
PY
done
