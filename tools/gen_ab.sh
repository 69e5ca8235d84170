#!/bin/bash
# Dev tool (GPU box): generation rate (bench.py's 1024-step resident run) for several library variants, like tools/ab.sh:
#   tools/gen_ab.sh old base old base   ("base" = the in-tree library; others music-generator_amd/lib/libdeepj_hip.<name>.so)
for v in "$@"; do
  if [ "$v" = base ]; then unset DEEPJ_LIB; else export DEEPJ_LIB=$PWD/music-generator_amd/lib/libdeepj_hip.$v.so; fi
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --no-fp32 --scaled-steps 0 > gpurun_out/genab_$v.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/genab_$v.log; exit 1; }
  python - $v <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/genab_{sys.argv[1]}.log") if l.startswith("{")][-1]); g = d["generation"]
print(sys.argv[1], "gen ms/time step", g["ms_per_time_step"], "notes/s", g["value"])
PY
done
