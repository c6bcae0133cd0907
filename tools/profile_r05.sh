#!/bin/bash
# Round-5 evidence, one gpurun call:  COMMIT=<short hash> bash tools/profile_r05.sh
#   for configs[1] (CartPole [64,64]) and configs[2] (Pendulum [256,256] + NormalizeWrapperEnv): rocprofv3 --kernel-trace --stats of the bench workload, then SEPARATE
#   --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ sets) over one epoch of it — never combined with a trace (gpurun refuses that), the program itself after `--`.
#   Output: gpurun_out/r05prof/<cfg>/{kernel_stats.csv, pmc_summary.json, grad_pmc.json}; grad_pmc.json names the commit and is what bench.py replays as roofline.traffic.
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r05prof; mkdir -p $OUT
COMMIT=${COMMIT:-unknown}
cd /tmp && export TMPDIR=/tmp
for CFG in ${CFGS:-cfg1 cfg2}; do
  if [ $CFG == cfg1 ]; then W=""; SAMPLES=4194304; else W="--env pendulum --hidden 256 --normalize"; SAMPLES=4194304; fi
  O=$OUT/$CFG; mkdir -p $O
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $W --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/trace_bench.json 2> $O/trace.err || echo "$CFG trace failed"
  ARGS="$W --steps 1 --warmup 0 --epochs 1 --no-cpu-baseline --no-events --no-secondary"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.json 2> $O/fetch.err || echo "$CFG fetch failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.json 2> $O/write.err || echo "$CFG write failed"
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY --output-format csv -d $O/sq1 -- python3 $R/bench.py $ARGS > $O/sq1.json 2> $O/sq1.err || echo "$CFG sq1 failed"
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $O/sq2 -- python3 $R/bench.py $ARGS > $O/sq2.json 2> $O/sq2.err || echo "$CFG sq2 failed"
  python3 - <<PY
import csv, glob, collections, json
O, commit, cfg = "$O", "$COMMIT", "$CFG"
out = {}
for d in ("fetch", "write", "sq1", "sq2"):
    for f in glob.glob(O + "/" + d + "/*/*_counter_collection.csv"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(k, {}).update({c: {"mean_per_launch": sum(x) / len(x), "launches": len(x)} for c, x in v.items()})
json.dump({"commit": commit, "workload": cfg, "kernels": out}, open(O + "/pmc_summary.json", "w"), indent=1)
stats = {}
for f in glob.glob(O + "/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0]] = (float(r["AverageNs"]) * 1e-6, int(r["Calls"]))
def rec_for(match):
    ks = [k for k in out if match in k]
    if not ks: return None
    g = ks[0]; rec = {"commit": commit, "workload": cfg, "kernel": g}
    for c, x in out[g].items(): rec[c + "_per_launch"] = x["mean_per_launch"]
    for k, (ms, calls) in stats.items():
        if match in k: rec["rocprof_avg_launch_ms"] = ms; rec["rocprof_calls"] = calls
    if "FETCH_SIZE_per_launch" in rec and "WRITE_SIZE_per_launch" in rec:
        # rocprofv3 reports KiB; gfx950 FETCH_SIZE tallies 64 B per 128-B request of wide streaming reads (MI355X_MICROARCH.md, HBM): x2 is the prescribed correction
        # (an upper bound for 32-byte record gathers, the raw figure the lower bound); WRITE_SIZE is exact for 16-B-per-lane stores
        rec["hbm_bytes_per_launch"] = (2 * rec["FETCH_SIZE_per_launch"] + rec["WRITE_SIZE_per_launch"]) * 1024
        rec["hbm_bytes_per_launch_uncorrected"] = (rec["FETCH_SIZE_per_launch"] + rec["WRITE_SIZE_per_launch"]) * 1024
    return rec
g = rec_for("ppo_grad")
if g:
    g["algorithmic_bytes_per_launch"] = $SAMPLES * 2 * 32
    g["note"] = "B = $SAMPLES samples per launch; both nets read one 32-byte record per sample"
    json.dump(g, open(O + "/grad_pmc.json", "w"), indent=1); print(json.dumps(g, indent=1))
for name in ("rollout_kernel", "gae_scan_kernel", "pack_records_kernel", "epoch_index_kernel", "epoch_moments_kernel", "grad_reduce_kernel", "adam_kernel"):
    r = rec_for(name)
    if r: json.dump(r, open(O + "/" + name + "_pmc.json", "w"), indent=1)
PY
  # kernel_stats.csv with the kernel names cut at the argument list (rocprofv3 prints full C++ signatures: `cut -c1-200` used to truncate the ROW of long ones — the
  # gae_scan_kernel line of profiles/r04_*/kernel_stats.csv lost its duration columns that way)
  python3 - <<PY
import csv, glob
rows = []
for f in glob.glob("$O/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        r["Name"] = r["Name"].split("(")[0].replace("void ", "").strip()
        rows.append(r)
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
if rows:
    w = csv.DictWriter(open("$O/kernel_stats.csv", "w", newline=""), fieldnames=list(rows[0].keys())); w.writeheader(); w.writerows(rows)
for r in rows[:12]: print(r["Name"][:60], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
done
