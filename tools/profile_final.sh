#!/bin/bash
# Round-1 evidence: kernel-trace stats of the default bench command + separate PMC passes (FETCH_SIZE / WRITE_SIZE) for the dominant kernel.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/final; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err || echo "trace failed"
ARGS="--steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err || echo "write failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- python3 $R/bench.py $ARGS > $OUT/sq1.json 2> $OUT/sq1.err || echo "sq1 failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.json 2> $OUT/sq2.err || echo "sq2 failed"
python3 - <<PY
import csv, glob, collections, json
out={}
for d in ("fetch","write","sq1","sq2"):
    for f in glob.glob("$OUT/"+d+"/*/*_counter_collection.csv"):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            out.setdefault(k,{}).update({c:{"mean_per_launch":sum(x)/len(x),"launches":len(x)} for c,x in v.items()})
json.dump(out, open("$OUT/pmc_summary.json","w"), indent=1)
g=[k for k in out if "ppo_grad" in k][0]
fetch=out[g]["FETCH_SIZE"]["mean_per_launch"]; write=out[g]["WRITE_SIZE"]["mean_per_launch"]
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads
# (MI355X_MICROARCH.md §HBM): the x2 correction is an UPPER bound for this kernel's 16-byte-per-lane gathers, raw value is the lower bound
json.dump({"kernel": g, "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
           "hbm_bytes_per_launch": (2*fetch + write)*1024, "hbm_bytes_per_launch_uncorrected": (fetch+write)*1024,
           "algorithmic_bytes_per_launch": 4194304*2*32, "note": "B = 4 194 304 samples per launch; both nets read one 32-B record per sample"},
          open("$OUT/ppo_grad_pmc.json","w"), indent=1)
print(open("$OUT/ppo_grad_pmc.json").read())
PY
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-160
