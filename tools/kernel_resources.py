#!/usr/bin/env python3
"""Static resource table of every kernel (VGPR / AGPR / SGPR / scratch / occupancy / LDS) from the compiler's own remarks.
No GPU needed:
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -c suitesparse_amd/csrc/engine.hip -o /tmp/e.o \
        -Rpass-analysis=kernel-resource-usage 2> /tmp/res.txt
  python tools/kernel_resources.py /tmp/res.txt profiles/rNN_kernel_resources.txt"""
import re, subprocess, sys
txt=open(sys.argv[1]).read()
blocks=re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
out=[]
out.append("# Static resource usage of every kernel in libmi355_cholmod.so (hipcc --offload-arch=gfx950 -O3, ROCm 7.2.0,")
out.append("# -Rpass-analysis=kernel-resource-usage; round 3).  No GPU is needed for this table.")
out.append(f"{'kernel':72s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'scratch B/lane':>15s} {'waves/SIMD':>11s} {'LDS B/WG':>9s}")
K=[r"VGPRs", r"AGPRs", r"SGPRs", r"ScratchSize \[bytes/lane\]", r"Occupancy \[waves/SIMD\]", r"LDS Size \[bytes/block\]"]
for b in blocks:
    name=b.split("\n")[0].split(" [-R")[0].strip()
    v=[]
    for k in K:
        m=re.search(k+r": (\d+)", b); v.append(int(m.group(1)) if m else -1)
    d=subprocess.run(["c++filt", name],capture_output=True,text=True).stdout.strip()
    d=re.sub(r"^void ","",d); d=d.replace("(anonymous namespace)::","")
    d=re.sub(r"\(.*","",d)
    out.append(f"{d[:72]:72s} {v[0]:5d} {v[1]:5d} {v[2]:5d} {v[3]:15d} {v[4]:11d} {v[5]:9d}")
open(sys.argv[2],"w").write("\n".join(out)+"\n")
print("\n".join(out))
