#!/bin/bash
# usage: tools/probe_env.sh VAR v1 v2 ... ; prints ms/step and one kernel category per value
var=$1; shift; cat_=${CAT:-head_loss}
for g in "$@"; do
  env $var=$g timeout -k 10 200 python bench.py --steps 3 --warmup 1 --cpu-sample 0 2>/dev/null > /tmp/probe.json
  python - "$var" "$g" "$cat_" <<'PY'
import json,sys
d=json.load(open("/tmp/probe.json")); print(sys.argv[1], sys.argv[2], "ms/step", d["ms_per_step"], sys.argv[3], d["kernel_ms_per_step"][sys.argv[3]])
PY
done
