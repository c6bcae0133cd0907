#!/bin/bash
# per-dispatch timeline of configs[4] (SAC): rocprofv3 --kernel-trace of a short run, then the dispatches of ONE steady-state iteration in order with grid sizes and durations
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-sac_trace}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --algo sac --steps 1 --warmup 1 --sac-iters 40 --no-cpu-baseline > $OUT/bench.json 2> $OUT/trace.err || echo "trace failed"
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# the last occurrence of the gather kernel starts the last complete update; print from the collection before it to the end
idx = [i for i, n in enumerate(names) if "sac_gather" in n]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
with open("$OUT/iteration.txt", "w") as o:
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("dril::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "").replace("dril::", "").split("(")[0].split("<")[0]
        line = f"{(s - t0) / 1e3:8.1f} us  +{(e - s) / 1e3:6.1f} us  grid {r['Grid_Size_X']:>7}x{r['Grid_Size_Y']:>4}x{r['Grid_Size_Z']:>2} wg {r['Workgroup_Size_X']:>4}  lds {r.get('LDS_Block_Size', '?'):>6}  {n}"
        print(line); o.write(line + "\n")
    tot = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
    print(f"iteration: {tot:.1f} us"); o.write(f"iteration: {tot:.1f} us\n")
PY
cp $OUT/trace/*/*_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
