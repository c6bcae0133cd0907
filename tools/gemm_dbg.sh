#!/bin/bash
# per-phase ablation of the LDS-tiled bf16-split contraction (DRIL_GEMM_DBG bits; results are wrong on purpose): kernel averages from rocprofv3
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 8 15; do
  export DRIL_DEBUG=1 DRIL_GEMM_DBG=$d GENERIC_SHAPE=${GENERIC_SHAPE:-64,18,1,512,512}
  rm -rf $R/gpurun_out/prof_dbg; rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dbg -- python3 $R/tools/generic_update.py > /dev/null 2>&1
  f=$(find $R/gpurun_out/prof_dbg -name "*kernel_stats.csv" | head -1)
  echo "DBG=$d"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'split_kernel' in r['Name']: print('   %-60s calls %6s avg %9.1f us' % (r['Name'].split('::')[-1][:60], r['Calls'], float(r['AverageNs'])/1e3))
"
done
