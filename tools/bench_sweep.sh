#!/bin/bash
# A/B sweep of one environment knob over the default bench (GPU box): tools/bench_sweep.sh VAR v1 v2 ...  -> one line per value
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python3 bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null > /tmp/sweep.json
  python3 - "$VAR=$v" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json")); r = d["roofline"]
print(sys.argv[1], "env-steps/s %.4g" % d["value"], "ms/step %.1f" % d["ms_per_step"], r["kernel"], "TFLOP/s %.2f" % r["achieved"], "launch ms %.4f" % r["avg_launch_ms"])
PY
done
