#!/bin/bash
# SQ counter sweep of the dominant kernel of the default bench workload (diagnostic; separate --pmc passes, no tracing domains)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/sweep; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events ${BENCH_EXTRA}"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_IFETCH"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py $ARGS > $OUT/g$i.json 2> $OUT/g$i.err || echo "group $i failed"
done
python3 - <<PY
import csv, glob, collections, json
out={}
for f in glob.glob("$OUT/g*/*/*_counter_collection.csv"):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items():
        out.setdefault(k,{}).update({c:sum(x)/len(x) for c,x in v.items()})
g=[k for k in out if "ppo_grad" in k][0]
json.dump({g: out[g]}, open("$OUT/sweep.json","w"), indent=1)
for c,x in sorted(out[g].items()): print("%-32s %.4g" % (c, x))
PY
