#!/bin/bash
# Register / spill / scratch report of every kernel in one HIP source (the compiler's view; needs no GPU):
#   tools/kernel_resources.sh agf_conv2d.hip [regex]
src=$1; filt=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -munsafe-fp-atomics -x hip --cuda-device-only \
  -c "$(dirname "$0")/../animeface_amd/csrc/$src" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
cur = None
def flush(c):
    if c: print("%-90s vgpr %3s agpr %3s scratch %4s spill %3s occ %s" % (c["name"][:90], c.get("VGPRs"), c.get("AGPRs"), c.get("ScratchSize"), c.get("VGPRs Spill"), c.get("Occupancy")))
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    k, _, v = t.partition(": ")
    if k == "Function Name":
        flush(cur); cur = {"name": v}
    elif cur is not None:
        cur[k.split(" [")[0]] = v
flush(cur)
' | grep -E "$filt"
