"""Register / scratch usage of every kernel of one source file (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
    python scripts/kernel_resources.py pokerrl_amd/csrc/prl_fhp_kernels.hip [filter] [-DDEFINE ...]"""
import os
import re
import subprocess
import sys

ROOT = os.environ.get("PRL_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = next((a for a in sys.argv[2:] if not a.startswith("-")), "")
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "pokerrl_amd", "csrc"),
       "--offload-arch=gfx950", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-rdc", "-x", "hip", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/lane\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
if not rows:
    sys.exit("no kernels found; compiler said:\n" + out[-3000:])
names = subprocess.run(["c++filt"] + list(rows), stdout=subprocess.PIPE, text=True).stdout.splitlines()
for mangled, name in zip(rows, names):
    if flt in name:
        r = rows[mangled]
        print("%-90s VGPR %3d  SGPR spill %3d  VGPR spill %3d  scratch %4d  LDS %6d  occupancy %d" % (
            name[:90], r.get("VGPRs", -1), r.get("SGPRs Spill", -1), r.get("VGPRs Spill", -1), r.get("ScratchSize", -1), r.get("LDS Size", -1), r.get("Occupancy", -1)))
