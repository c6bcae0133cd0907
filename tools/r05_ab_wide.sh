#!/bin/bash
# Round 5, one gpurun call: the GPU test suite, then a same-box A/B of configs[2] (Pendulum [256,256] + NormalizeWrapperEnv) between
#   ab_r3/  = the tree at 2e39a12 (round-3 HEAD) with its own library and bench.py
#   ab_r4/  = libdril_hip.so built at 33d0f8c (round-4 HEAD), under this tree's bench.py (DRIL_HIP_LIBRARY)
#   HEAD    = this tree
# (how the two trees were made: `mkdir ab_r3 && git archive 2e39a12 | tar -x -C ab_r3 && make -C ab_r3/dril.jl_amd/csrc libdril_hip.so`; `mkdir ab_r4 && git archive 33d0f8c dril.jl_amd/csrc include |
#  tar -x -C ab_r4 && make -C ab_r4/dril.jl_amd/csrc libdril_hip.so`; both git-ignored, travelling to the box with the gpurun snapshot; removed after the round's A/Bs — absent trees are skipped)
# alternating, ROUNDS times.  Answers VERDICT r4 item 1(a) (did ppo_grad_wide_split_kernel regress between rounds 3 and 4?) and measures the round-5 kernel beside both.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${TAG:-r05_ab}; mkdir -p $OUT; ROUNDS=${ROUNDS:-3}
cd $R
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS} > $OUT/pytest.log 2>&1; rc=$?; tail -5 $OUT/pytest.log; echo "pytest rc $rc"
  if [ $rc -ge 124 ]; then echo "pytest timed out / was killed: no further GPU step"; exit $rc; fi
fi
W="--env pendulum --hidden ${HIDDEN:-256} --normalize --steps 2 --warmup 1 --no-secondary --no-cpu-baseline"
summ() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('$1', 'value %.4g' % d['value'], 'ms/step %.1f' % d['ms_per_step'], r.get('kernel'), 'launch ms %.4f' % r.get('avg_launch_ms', float('nan')), 'TFLOP/s %.1f' % r.get('achieved', float('nan')))"; }
for i in $(seq $ROUNDS); do
  if [ -d ab_r3 ] && [ -z "$SKIP_R3" ]; then (cd ab_r3 && timeout -k 10 300 python3 bench.py $W 2>$OUT/r3_$i.err | tail -1 | tee $OUT/r3_$i.json | summ r3_2e39a12) || { echo "r3 run failed"; tail -3 $OUT/r3_$i.err; }; fi
  if [ -d ab_r4 ]; then (DRIL_HIP_LIBRARY=$R/ab_r4/dril.jl_amd/csrc/libdril_hip.so timeout -k 10 300 python3 bench.py $W 2>$OUT/r4_$i.err | tail -1 | tee $OUT/r4_$i.json | summ r4_33d0f8c) || { echo "r4 run failed"; tail -3 $OUT/r4_$i.err; }; fi
  (timeout -k 10 300 python3 bench.py $W 2>$OUT/head_$i.err | tail -1 | tee $OUT/head_$i.json | summ HEAD) || { echo "HEAD run failed"; tail -3 $OUT/head_$i.err; }
done
