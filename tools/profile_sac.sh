#!/bin/bash
# rocprofv3 kernel trace of the SAC bench (run on the GPU box via gpurun from the repo root)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_sac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --algo sac --steps 1 --warmup 1 --sac-iters ${SAC_ITERS:-100} --no-cpu-baseline --no-events > $OUT/trace_bench.json 2> $OUT/trace.err || echo "trace pass failed"
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
head -30 $OUT/kernel_stats.csv
