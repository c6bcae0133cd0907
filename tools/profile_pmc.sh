#!/bin/bash
# PMC-only passes (separate runs per counter group); usage: tools/profile_pmc.sh <tag>
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-pmc}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- python3 $R/bench.py $ARGS > $OUT/sq1.json 2> $OUT/sq1.err || echo "sq1 failed"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.json 2> $OUT/sq2.err || echo "sq2 failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $OUT/sq3 -- python3 $R/bench.py $ARGS > $OUT/sq3.json 2> $OUT/sq3.err || echo "sq3 failed"
python3 - <<PY
import csv, glob, collections, json
out={}
for d in ("sq1","sq2","sq3"):
    for f in glob.glob("$OUT/"+d+"/*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            out.setdefault(k,{}).update({c:{"mean_per_launch":sum(x)/len(x),"launches":len(x)} for c,x in v.items()})
json.dump(out, open("$OUT/pmc_summary.json","w"), indent=1)
for k in out:
    if "ppo_grad" in k or "rollout" in k: print(k, {c:round(v["mean_per_launch"]) for c,v in out[k].items()})
PY
