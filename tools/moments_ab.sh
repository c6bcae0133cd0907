#!/bin/bash
# timing of epoch_moments_kernel / its finalize inside one bench iteration (rocprofv3 kernel stats): bash tools/moments_ab.sh  (on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
for blocks in 2048 512 4096; do
  export DRIL_MOMENT_BLOCKS=$blocks
  d=gpurun_out/mom2_$blocks
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $d.log 2>&1 || exit 1
  echo "blocks $blocks: $(grep -h epoch_moments $d/p_kernel_stats.csv $d/*/p_kernel_stats.csv 2>/dev/null | cut -d, -f1-4 | tr '\n' ' ')"
  tail -1 $d.log | cut -c1-200
done
