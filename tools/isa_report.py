#!/usr/bin/env python3
"""ISA report of the solve kernel: registers, spills, and -- what matters for the LDS-DMA prefetch -- every scratch
reload / vmcnt wait with its loop depth (a reload between a prefetch and its use makes the prefetch synchronous,
because vmcnt completes in order).   python tools/isa_report.py [--all]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp()
s_path = os.path.join(d, "solver.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17", "-I" + ROOT + "/include",
                       "-I" + ROOT + "/carnd-mpc-project_amd/csrc", "--cuda-device-only", "-S", "-o", s_path,
                       ROOT + "/carnd-mpc-project_amd/csrc/mpc_solver.hip"] + os.environ.get("EXTRA", "").split(), stderr=subprocess.DEVNULL)
txt = open(s_path).read()
KERNEL = os.environ.get("KERNEL", "_ZN12_GLOBAL__N_116mpc_solve_kernelILb1EdLi1E")   # <STAGING=true, double, OCC=1>; fp32: ...ILb1EfLi2E
m = re.search(r"^" + KERNEL + r".*?s_endpgm", txt, re.S | re.M)
k = m.group(0).split("\n")
meta = re.findall(r"; (NumVgprs|NumAgprs|TotalNumVgprs|ScratchSize|codeLenInByte|Occupancy)[:=]? *=? *(\d+)", txt[m.end():m.end() + 12000])
print(KERNEL, dict(meta[:6]))
depth = 0; label = ""; cnt = 0; out = []
stats = {"scratch_load": 0, "scratch_store": 0, "valu": 0, "accvgpr": 0, "dma": 0}
deep = 0
for i, l in enumerate(k):
    mm = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
    if mm:
        dd = re.search(r"Depth=(\d+)", l); depth = int(dd.group(1)) if dd else 0; label = mm.group(1)
    s = l.strip()
    if not s or s[0] in ";.": continue
    op = s.split()[0]
    if op.startswith("v_"): stats["valu"] += 1
    if "accvgpr" in op: stats["accvgpr"] += 1
    if "global_load_lds" in op: stats["dma"] += 1; cnt += 1; continue
    if op.startswith("scratch_load"):
        stats["scratch_load"] += 1
        if depth >= 3: deep += 1
    if op.startswith("scratch_store"): stats["scratch_store"] += 1
    if op.startswith("scratch_") or op.startswith("s_waitcnt") and "vmcnt" in s:
        if cnt: out.append("        ... %d x DMA [d%d]" % (cnt, depth)); cnt = 0
        out.append("%6d [%s d%d] %s" % (i, label, depth, s[:90]))
print(stats, "scratch reloads at loop depth >= 3:", deep)
if "--all" in sys.argv: print("\n".join(out))
else: print("\n".join(o for o in out if re.search(r"d[3-9]\]", o)))
