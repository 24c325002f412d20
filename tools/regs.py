"""VGPR / scratch / occupancy per kernel:  python tools/regs.py <file.hip> <regex> [-DFLAG ...]"""
import re, subprocess, sys, os
src, pat, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "electrocardio_panorama_amd", "csrc")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", os.path.join(csrc, "..", "..", "include"),
       "-I", csrc, *flags, "-c", os.path.join(csrc, src), "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for line in out.splitlines():
    m = re.search(r"remark:\s+([^:]+): (.*) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur[k] = v
names = [r["name"] for r in rows]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
for r, d in zip(rows, dem):
    d = re.sub(r"^void \(anonymous namespace\)::", "", d)
    d = re.sub(r"\(.*", "", d)
    if re.search(pat, d):
        print(f"{d:60s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')} lds {r.get('LDS Size [bytes/block]','?')}")
