#!/bin/bash
# rocprofv3 kernel trace of a short config-3 run (Pendulum, [256,256], NormalizeWrapperEnv)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/prof_cfg3; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --env pendulum --hidden 256 --normalize --n-steps 128 --minibatches 4 --epochs 1 --steps 1 --warmup 1 --no-cpu-baseline --no-events > $OUT/bench.json 2> $OUT/err.txt || echo failed
f=$(ls -t $OUT/trace/*/*_kernel_stats.csv | head -1); cp $f $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv | cut -c1-170
