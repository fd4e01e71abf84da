#!/usr/bin/env python3
"""Condenses the compiler's resource remarks (make -C pytorch-retinanet_amd/csrc asm ->
build/*.usage.txt) into one line per kernel: usage tools/kernel_resources.py > profiles/rNN_kernel_resources.txt"""
import glob, os, re, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kernels = []
for f in sorted(glob.glob(os.path.join(ROOT, "pytorch-retinanet_amd", "csrc", "build", "*.usage.txt"))):
    cur = {}
    for line in open(f):
        m = re.search(r"remark: (.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            kernels.append(cur)
        elif ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in kernels), capture_output=True, text=True).stdout.split("\n")
print("# Register use of every kernel: hipcc -Rpass-analysis=kernel-resource-usage (make -C pytorch-retinanet_amd/csrc asm), gfx950")
print("# kernel | VGPRs | AGPRs | SGPRs | scratch bytes/lane | VGPR spills | SGPR spills | LDS bytes/block | waves/SIMD")
for k, nm in zip(kernels, names):
    nm = re.sub(r"\(.*", "", nm)
    print(f"{nm} | {k.get('VGPRs')} | {k.get('AGPRs')} | {k.get('TotalSGPRs')} | {k.get('ScratchSize [bytes/lane]')} | "
          f"{k.get('VGPRs Spill')} | {k.get('SGPRs Spill')} | {k.get('LDS Size [bytes/block]')} | {k.get('Occupancy [waves/SIMD]')}")
