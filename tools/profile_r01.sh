#!/bin/bash
# rocprofv3 passes for round 1 (run on the GPU box via gpurun from the repo root). Kernel trace and PMC passes are separate runs.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-events > $OUT/trace_bench.json 2> $OUT/trace.err || echo "trace pass failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err || echo "pmc sq pass failed"
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_sq2 -- python3 $R/bench.py --steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events > $OUT/pmc_sq2.json 2> $OUT/pmc_sq2.err || echo "pmc sq2 pass failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err || echo "pmc fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events > $OUT/pmc_write.json 2> $OUT/pmc_write.err || echo "pmc write pass failed"
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
ls -R $OUT | head -50
