#!/bin/bash
# same-box A/B of two builds of the library on a bench configuration: tools/ab_lib.sh "<bench.py args>" libA.so libB.so [rounds]  (paths relative to dril.jl_amd/csrc)
ARGS=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do for lib in $A $B; do
  DRIL_HIP_LIBRARY=$PWD/dril.jl_amd/csrc/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', '%.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], r.get('kernel'), 'launch ms %.4f' % r['avg_launch_ms'], 'TFLOP/s %.2f' % r['achieved'])"
done; done
