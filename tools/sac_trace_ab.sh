#!/bin/bash
# same-box A/B of one SAC switch: per-dispatch timelines (tools/sac_trace.sh) and un-traced bench lines with the variable unset / set.  usage: sac_trace_ab.sh DRIL_SAC_NO_FUSED_DW1
V=${1:?variable}; R=${GRAFT_REPO_ROOT:-$PWD}
bash $R/tools/sac_trace.sh ab_on > /dev/null 2>&1 && cat $R/gpurun_out/ab_on/iteration.txt && env $V=1 bash $R/tools/sac_trace.sh ab_off > /dev/null 2>&1 && cat $R/gpurun_out/ab_off/iteration.txt || exit 1
for i in 1 2; do
  python3 $R/bench.py --algo sac --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('default ', '%.4g' % d['value'], d['roofline'].get('update_ms'), d['roofline'].get('collect_step_ms'))" || exit 1
  env $V=1 python3 $R/bench.py --algo sac --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=1', '%.4g' % d['value'], d['roofline'].get('update_ms'), d['roofline'].get('collect_step_ms'))" || exit 1
done
