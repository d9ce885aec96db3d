"""registers, spills and occupancy of the render kernels of the development build: python scripts/kernel_resources.py [-D flags]"""
import os, re, subprocess, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-volpath_amd")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
       "-fno-fast-math", "-fno-slp-vectorize", "-DVP_DEV_BUILD", *sys.argv[1:], "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
       "-c", "csrc/vp_kernels.hip", "-o", "/tmp/kres.o"]
out = subprocess.run(cmd, cwd=root, capture_output=True, text=True).stderr
cur, rows = None, []
for l in out.splitlines():
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", l)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    if "render_k" not in r["name"] and "approach" not in r["name"]:
        continue
    n = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    n = re.sub(r"\(SceneDev, LaunchDev\)|void ", "", n.replace("vp::", ""))
    print(f"{n:58s} VGPR {r.get('VGPRs'):>4} SGPR {r.get('TotalSGPRs'):>4} spill s/v {r.get('SGPRs Spill')}/{r.get('VGPRs Spill')} occ {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')}")
