#!/bin/bash
# Register / spill / LDS table of every kernel of a translation unit (compile only; runs without a GPU):
#   scripts/kernel_resources.sh filter_f16 [extra hipcc flags]
cd "$(dirname "$0")/../fast-match_amd/csrc"
U=$1; shift
X=""; [ "$U" = dist_f32 ] && X=-fno-slp-vectorize
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $X "$@" -Rpass-analysis=kernel-resource-usage -c $U.hip -o /tmp/res_$U.o 2>&1 | python3 -c '
import sys, re
rows = []; cur = None
for line in sys.stdin:
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]|TotalSGPRs): (.*?) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None: cur[k] = v
import subprocess
print("%-72s %5s %5s %5s %7s %7s %5s %7s" % ("kernel", "VGPR", "AGPR", "SGPR", "spillV", "scratch", "occ", "LDS"))
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"\(.*$", "", name).replace("void ", "")
    print("%-72s %5s %5s %5s %7s %7s %5s %7s" % (name[:72], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
