#!/usr/bin/env python3
"""Register / LDS / scratch / occupancy of every kernel as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py [file.hip ...]   (default: every .hip under gnss-dsp-tools_amd/csrc)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gnss-dsp-tools_amd", "csrc")


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    for f in files:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I/opt/rocm/include",
                            "-Rpass-analysis=kernel-resource-usage", "-c", f, "-o", "/dev/null"], capture_output=True, text=True)
        blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
        print("== %s" % os.path.basename(f))
        for b in blocks:
            name = b.split("\n")[0].strip()
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            dn = re.sub(r"\(.*", "", dn)[:80]
            g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
            print("%-80s vgpr %3s agpr %3s sgpr %3s scratch %3s occ %s lds %6s" % (
                dn, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"),
                g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main()
