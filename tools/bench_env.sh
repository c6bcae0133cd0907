#!/bin/bash
# A/B runs of a bench configuration on ONE GPU box (boxes differ by +-3 %): tools/bench_env.sh [--args "<bench.py args>"] "A=1,B=2" "A=3" ...  -> one line per environment set ("-" = defaults)
ARGS=""; if [ "$1" == "--args" ]; then ARGS="$2"; shift 2; fi
for spec in "$@"; do
  envs=(); if [ "$spec" != "-" ]; then IFS=',' read -ra envs <<< "$spec"; fi
  env "${envs[@]}" python3 bench.py --no-secondary --no-cpu-baseline --steps 2 --warmup 1 $ARGS 2>/dev/null > /tmp/sweep.json
  python3 - "$spec" <<'PY'
import json, sys
d = json.load(open("/tmp/sweep.json")); r = d["roofline"]
print(sys.argv[1], "env-steps/s %.4g" % d["value"], "ms/step %.1f" % d["ms_per_step"], r["kernel"], "TFLOP/s %.2f" % r["achieved"], "launch ms %.4f" % r["avg_launch_ms"])
PY
done
