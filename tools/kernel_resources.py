#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / code-size table of one translation unit of libdril_hip.so (cross-compiles; no GPU needed).

usage: tools/kernel_resources.py dril_grad_wide.hip [name-filter] [extra hipcc flags ...]
Reads hipcc's -Rpass-analysis=kernel-resource-usage remarks; code size per kernel = the symbol's size in the gfx950 code object."""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dril.jl_amd", "csrc")
flags = "-O3 -fno-slp-vectorize -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950".split() + ([] if "NO_VGPR_FORM" in os.environ else ["-mllvm", "-amdgpu-mfma-vgpr-form"])
with tempfile.TemporaryDirectory() as td:
    obj = os.path.join(td, "o.o")
    r = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *extra, "-Rpass-analysis=kernel-resource-usage", "-save-temps=obj", "-c", os.path.join(csrc, src), "-o", obj], capture_output=True, text=True, cwd=csrc)
    if r.returncode: sys.exit(r.stderr[-3000:])
    sizes = {}
    for f in os.listdir(td):
        if f.endswith(".out") or "amdgcn" in f and f.endswith(".o"):
            s = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-sW", os.path.join(td, f)], capture_output=True, text=True).stdout
            for ln in s.splitlines():
                p = ln.split()
                if len(p) >= 8 and p[3] == "FUNC": sizes[p[7]] = int(p[2])
rows, cur = [], None
for ln in r.stderr.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass-analysis", ln)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"): cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ":" in t: k, v = t.split(":", 1); cur[k.strip()] = v.strip()
def dem(n):
    return subprocess.run(["/usr/bin/c++filt", n], capture_output=True, text=True).stdout.strip().replace("dril::", "").split("(")[0]
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'vspill':>6s} {'sspill':>6s} {'scratch':>7s} {'occ':>3s} {'LDS':>7s} {'code B':>8s}")
for c in rows:
    n = dem(c["name"])
    if filt and filt not in n: continue
    print(f"{n[:70]:70s} {c.get('VGPRs','?'):>5s} {c.get('AGPRs','?'):>5s} {c.get('VGPRs Spill','?'):>6s} {c.get('SGPRs Spill','?'):>6s} {c.get('ScratchSize [bytes/lane]','?'):>7s} {c.get('Occupancy [waves/SIMD]','?'):>3s} {c.get('LDS Size [bytes/block]','?'):>7s} {sizes.get(c['name'], 0):>8d}")
