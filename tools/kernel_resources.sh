#!/bin/bash
# Per-kernel register / LDS / scratch usage of one translation unit (device pass only):
#   tools/kernel_resources.sh ss_asr_amd/csrc/decoder.hip [pattern]
# Prints "kernel  VGPRs  AGPRs  spill  scratch  LDS  occupancy" for kernels matching pattern.
src=$1; pat=${2:-.}
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-pass-failed --cuda-device-only -c "$src" -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark: (.*)", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].split("[-R")[0].strip()}
        rows.append(cur)
    elif ":" in t and cur is not None:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.split("[-R")[0].strip()
import subprocess
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    if not re.search(sys.argv[1], name): continue
    print("%-70s V %3s A %3s spill %s scratch %s LDS %s occ %s" % (name[:70], r.get("VGPRs"), r.get("AGPRs"),
          r.get("VGPRs Spill", r.get("VGPR Spill")), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "$pat"
