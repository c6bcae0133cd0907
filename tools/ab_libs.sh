#!/bin/bash
# same-box A/B of any number of builds of the library: tools/ab_libs.sh "<bench.py args>" ROUNDS lib1.so lib2.so ...   (paths relative to dril.jl_amd/csrc; alternating)
ARGS=$1; N=$2; shift 2
for i in $(seq $N); do for lib in "$@"; do
  DRIL_HIP_LIBRARY=$PWD/dril.jl_amd/csrc/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary $ARGS 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib', '%.4g' % d['value'], 'ms/step %.1f' % d['ms_per_step'], r.get('kernel'), 'launch ms %.4f' % r['avg_launch_ms'], 'TFLOP/s %.2f' % r['achieved'])"
done; done
