#!/bin/bash
# same-box A/B of two builds of the library (dril.jl_amd/csrc/libdril_old.so / libdril_new.so: `git stash; make; cp libdril_hip.so libdril_old.so; git stash pop; make; cp ... libdril_new.so`)
# on configs[1] with the rollout kernel's time beside the iteration's
for i in 1 2 3; do for lib in libdril_old.so libdril_new.so; do
  DRIL_HIP_LIBRARY=$PWD/dril.jl_amd/csrc/$lib timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', '%.4g' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'rollout ms %.3f' % d['kernel_ms_per_step']['rollout_kernel'])"
done; done
