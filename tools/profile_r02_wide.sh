#!/bin/bash
# Round-2 evidence for BASELINE configs[2] (Pendulum, [256,256], NormalizeWrapperEnv): bench line, rocprofv3 kernel stats, PMC passes of ppo_grad_wide_split_kernel
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r02w; mkdir -p $OUT
W="--env pendulum --hidden 256 --normalize"
cd $R
timeout -k 10 400 python3 bench.py $W --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_config3.json 2> $OUT/bench_config3.err || echo "bench failed"
DRIL_GRAD_VARIANT=0 timeout -k 10 400 python3 bench.py $W --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_config3_f32.json 2> $OUT/bench_config3_f32.err || echo "f32 bench failed"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $W --steps 1 --warmup 1 --no-cpu-baseline > $OUT/trace_bench.json 2> $OUT/trace.err || echo "trace failed"
ARGS="$W --steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err || echo "write failed"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- python3 $R/bench.py $ARGS > $OUT/sq1.json 2> $OUT/sq1.err || echo "sq1 failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.json 2> $OUT/sq2.err || echo "sq2 failed"
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
rec={"kernel": g}
for c,x in out[g].items(): rec[c+"_per_launch"]=x["mean_per_launch"]
for f in glob.glob("$OUT/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ppo_grad" in r["Name"]: rec["rocprof_avg_launch_ms"]=float(r["AverageNs"])*1e-6; rec["rocprof_calls"]=int(r["Calls"])
rec["hbm_bytes_per_launch"]=(2*rec["FETCH_SIZE_per_launch"]+rec["WRITE_SIZE_per_launch"])*1024
rec["hbm_bytes_per_launch_uncorrected"]=(rec["FETCH_SIZE_per_launch"]+rec["WRITE_SIZE_per_launch"])*1024
rec["algorithmic_bytes_per_launch"]=4194304*2*32
json.dump(rec, open("$OUT/wide_split_pmc.json","w"), indent=1)
print(json.dumps(rec, indent=1))
PY
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-170 > $OUT/kernel_stats.csv; head -8 $OUT/kernel_stats.csv
for f in bench_config3 bench_config3_f32; do python3 -c "
import json
d=json.load(open('$OUT/$f.json')); r=d['roofline']; print('$f', '%.4g'%d['value'], d['dtype'], r['kernel'], '%.1f'%r['achieved'], '%.3f'%r['frac'], d['ms_per_step'])"; done
