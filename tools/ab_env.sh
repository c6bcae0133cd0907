#!/bin/bash
# same-box A/B of one library under two environments: tools/ab_env.sh "<bench.py args>" ROUNDS "VAR=a" "VAR=b" ...   (alternating; an empty string = the default environment)
ARGS=$1; N=$2; shift 2
for i in $(seq $N); do for e in "$@"; do
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('[$e]', '%.4g' % d['value'], 'ms/step %.1f' % d['ms_per_step'], r.get('kernel'), 'launch ms %.4f' % r['avg_launch_ms'], 'TFLOP/s %.2f' % r['achieved'])"
done; done
