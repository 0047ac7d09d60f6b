#!/usr/bin/env python3
"""Registers / scratch / occupancy / code size of every instantiation of the solve and tail kernels (from the ISA metadata hipcc
emits for gfx950).  DESIGN.md section 4's table is this script's output.   python tools/isa_table.py"""
import os, re, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = tempfile.mkdtemp(); sp = os.path.join(d, "s.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17", "-I" + ROOT + "/include",
                       "-I" + ROOT + "/carnd-mpc-project_amd/csrc", "--cuda-device-only", "-S", "-o", sp,
                       ROOT + "/carnd-mpc-project_amd/csrc/mpc_solver.hip"], stderr=subprocess.DEVNULL)
txt = open(sp).read()
T = {"d": "double", "f": "float"}
print("| kernel <staging, solver reals, waves/SIMD, ABI reals[, source reals]> | VGPR + AGPR | scratch | waves/SIMD | code |")
print("|---|---|---|---|---|")
for m in re.finditer(r"^(_ZN12_GLOBAL__N_1\d+mpc_(solve|tail_slice)_kernelILb([01])E([df])Li(\d)E([df])([df])?EE\w*):", txt, re.M):
    K, kind, stg, r, occ, rio, rsrc = m.groups()
    mm = re.search(r"^" + re.escape(K) + r":.*?s_endpgm", txt, re.S | re.M)
    meta = dict(re.findall(r"; (NumVgprs|NumAgprs|ScratchSize|codeLenInByte|Occupancy)[:=]? *=? *(\d+)", txt[mm.end():mm.end() + 12000])[:5])
    name = "mpc_%s_kernel<%s, %s, %s, %s%s>" % (kind, "true" if stg == "1" else "false", T[r], occ, T[rio], ", " + T[rsrc] if rsrc else "")
    print("| `%s` | %s + %s = %d | %s B | %s | %.1f KB |" % (name, meta["NumVgprs"], meta["NumAgprs"], int(meta["NumVgprs"]) + int(meta["NumAgprs"]),
                                                          meta["ScratchSize"], meta["Occupancy"], int(meta["codeLenInByte"]) / 1024))
