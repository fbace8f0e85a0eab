#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr log) per kernel."""
import re
import subprocess
import sys


def main(path):
    s = open(path).read()
    blocks = re.split(r"remark: [^\n]*Function Name: ", s)[1:]
    for b in blocks:
        name = b.split("\n")[0].strip()

        def g(k):
            m = re.search(k + r": (\d+)", b)
            return int(m.group(1)) if m else -1

        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dn = re.sub(r"\(.*", "", dn).replace("void (anonymous namespace)::", "")
        print("%-52s vgpr=%4d agpr=%3d sgpr=%4d scratch=%5d occ=%d lds=%6d" % (
            dn, g(r"[^l]VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
            g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))


if __name__ == "__main__":
    main(sys.argv[1])
