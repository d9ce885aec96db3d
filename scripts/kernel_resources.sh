#!/bin/bash
# registers, spills and occupancy of the render kernels of the development build: scripts/kernel_resources.sh [extra -D flags]
cd "$(dirname "$0")/../cuda-volpath_amd"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize \
  -DVP_DEV_BUILD "$@" --cuda-device-only -Rpass-analysis=kernel-resource-usage -c csrc/vp_kernels.hip -o /tmp/kres.o 2>&1 | python3 -c '
import re,sys,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r"remark: (.*?): (.*?) \[-Rpass",l) or re.search(r"remark:\s+(.*?): (.*?) \[-Rpass",l)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=="Function Name": cur={"name":v}; rows.append(cur)
    elif cur is not None: cur[k]=v
for r in rows:
    if "render_k" not in r["name"]: continue
    n=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()
    n=re.sub(r"vp::|\(vp::SceneDev, vp::LaunchDev\)|void ","",n)
    print(f"{n:60s} VGPR {r.get(\"VGPRs\")}  SGPR {r.get(\"TotalSGPRs\")}  spill s/v {r.get(\"SGPRs Spill\")}/{r.get(\"VGPRs Spill\")}  occ {r.get(\"Occupancy [waves/SIMD]\")}  LDS {r.get(\"LDS Size [bytes/block]\")}")
'
