#!/bin/bash
# Round-3 evidence (run on the GPU box from the repo root: gpurun -- 'bash tools/profile_r03.sh'): the default bench line (with its secondary runs of
# configs[2] / configs[4] / configs[0]), rocprofv3 kernel-trace stats of the same workload, separate --pmc passes for the dominant kernel (HBM bytes;
# matrix-core op counters), kernel stats of the README-sized run (ppo_update_small_kernel) and of SAC.  Everything lands in gpurun_out/r03/; the
# summaries are copied into profiles/ by hand afterwards.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r03; mkdir -p $OUT
cd $R
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || echo "default bench failed"
echo "bench done"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/trace_bench.json 2> $OUT/trace.err || echo "trace failed"
echo "trace done"
ARGS="--steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events --no-secondary"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $ARGS > $OUT/fetch.json 2> $OUT/fetch.err || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $ARGS > $OUT/write.json 2> $OUT/write.err || echo "write failed"
echo "hbm pmc done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $OUT/sq1 -- python3 $R/bench.py $ARGS > $OUT/sq1.json 2> $OUT/sq1.err || echo "sq1 failed"
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $OUT/sq2 -- python3 $R/bench.py $ARGS > $OUT/sq2.json 2> $OUT/sq2.err || echo "sq2 failed"
echo "sq pmc done"
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
avg=None
for f in glob.glob("$OUT/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ppo_grad" in r["Name"]: avg=(float(r["AverageNs"])*1e-6, int(r["Calls"]))
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide streaming reads
# (MI355X_MICROARCH.md section HBM): the x2 correction is an UPPER bound for this kernel's 16-byte-per-lane gathers, the raw value the lower bound
rec={"kernel": g, "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
     "hbm_bytes_per_launch": (2*fetch + write)*1024, "hbm_bytes_per_launch_uncorrected": (fetch+write)*1024,
     "algorithmic_bytes_per_launch": 4194304*2*32, "note": "B = 4 194 304 samples per launch; both nets read one 32-B record per sample; a random 32-B gather moves at least one 64-B HBM burst, so 2x algorithmic is the floor of this access pattern",
     "rocprof_avg_launch_ms": avg[0] if avg else None, "rocprof_source": f"profiles/r03_final/kernel_stats.csv ({avg[1]} calls)" if avg else None}
for c in ("SQ_INSTS_VALU_MFMA_MOPS_F32","SQ_INSTS_VALU_MFMA_MOPS_BF16","SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CYCLES","SQ_WAVE_CYCLES","SQ_LDS_BANK_CONFLICT","SQ_INSTS_LDS","SQ_INSTS_VALU","GRBM_GUI_ACTIVE","SQ_ACTIVE_INST_VALU"):
    if c in out[g]: rec[c+"_per_launch"]=out[g][c]["mean_per_launch"]
json.dump(rec, open("$OUT/ppo_grad_pmc.json","w"), indent=1)
print(open("$OUT/ppo_grad_pmc.json").read())
PY
cat $OUT/trace/*/*_kernel_stats.csv | cut -c1-170 > $OUT/kernel_stats.csv; head -12 $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/smalltrace -- python3 $R/bench.py --n-envs 4 --minibatches 128 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/small_bench.json 2> $OUT/small.err || echo "small trace failed"
cat $OUT/smalltrace/*/*_kernel_stats.csv | cut -c1-170 > $OUT/small_kernel_stats.csv; head -8 $OUT/small_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sactrace -- python3 $R/bench.py --algo sac --steps 1 --warmup 1 --sac-iters 200 --no-cpu-baseline > $OUT/sactrace_bench.json 2> $OUT/sactrace.err || echo "sac trace failed"
cat $OUT/sactrace/*/*_kernel_stats.csv | cut -c1-170 > $OUT/sac_kernel_stats.csv
cd $R
timeout -k 10 300 python3 bench.py --env pendulum --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_pendulum64.json 2> $OUT/bench_pendulum64.err || echo "pendulum bench failed"
timeout -k 10 300 python3 bench.py --hidden 128 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_hidden128.json 2> $OUT/bench_hidden128.err || echo "hidden128 bench failed"
DRIL_GRAD_VARIANT=0 timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_f32_variant.json 2> $OUT/bench_f32_variant.err || echo "f32 variant bench failed"
for f in bench_default bench_pendulum64 bench_hidden128 bench_f32_variant small_bench; do python3 -c "
import json,sys
try:
    d=json.load(open('$OUT/$f.json')); r=d.get('roofline',{})
    print('$f', '%.4g'%d['value'], d['dtype'], r.get('kernel'), '%.1f'%r.get('achieved',0), '%.3f'%r.get('frac',0), '%.3f'%r.get('frac_vs_f32_peak',0))
    for x in d.get('secondary',[]): rr=x.get('roofline',{}); print('   secondary', x['config']['workload'][:60], '%.4g'%x['value'], rr.get('kernel','')[:30], '%.1f'%rr.get('achieved',0), '%.3f'%rr.get('frac',0), rr.get('update_ms'))
except Exception as e: print('$f', 'unreadable', e)
"; done
