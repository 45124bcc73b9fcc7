# usage: bash tools/regs.sh   -- VGPRs / spills / occupancy of every kernel in the device unit (compile only)
cd "$(dirname "$0")/../rsparse_amd/csrc" && /opt/rocm/bin/hipcc -std=c++17 -O3 -fPIC --offload-arch=gfx950 -Wno-unused-parameter \
  -Rpass-analysis=kernel-resource-usage -c device/device_unit.hip -o /tmp/regs_unit.o 2>&1 | python3 -c '
import sys,re
cur=None
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r"remark:\s+(VGPRs|AGPRs|VGPRs Spill|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)",line)
    if m and cur:
        d[m.group(1)]=m.group(2)
        if m.group(1).startswith("LDS"): print(cur[:90], d)
'
