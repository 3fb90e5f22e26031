#!/usr/bin/env python3
"""Static resources of every kernel as hipcc compiles it for gfx950 (no GPU needed): VGPRs, AGPRs, SGPRs, scratch (spill) bytes per
lane, static LDS bytes per workgroup, the compiler's occupancy figure (waves per SIMD), code bytes.  From `hipcc -S
--cuda-device-only` with the product's flags (compress_amd/build.py).

    python tools/isa_resources.py > profiles/rNN_isa_resources.txt
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "compress_amd", "csrc")
sys.path.insert(0, ROOT)
from compress_amd import build as kbuild  # noqa: E402


def main():
    rows = []
    with tempfile.TemporaryDirectory() as t:
        for f in sorted(x for x in os.listdir(CSRC) if x.endswith(".hip")):
            s = os.path.join(t, "k.s")
            subprocess.run([kbuild.HIPCC] + kbuild.FLAGS + ["-S", "--cuda-device-only", os.path.join(CSRC, f), "-o", s],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            txt = open(s).read()
            kernels = set(re.findall(r"^\s*\.amdhsa_kernel (\S+)", txt, re.M))
            for name in sorted(kernels, key=lambda k: txt.index("\t.type\t%s,@function" % k)):
                a = txt.index("\t.type\t%s,@function" % name)
                mo = re.compile(r"^; Occupancy: (\d+)", re.M).search(txt, a)
                body, occ = txt[a:mo.end()], mo.group(1)

                def g(key):
                    r = re.findall(r"^; %s[:=\s]+(\d+)" % key, body, re.M)
                    return r[-1] if r else "?"
                dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE).stdout.decode().strip()
                dem = re.sub(r"\(.*\)$", "", dem).replace("void ", "")
                rows.append((f, dem, g("NumVgprs"), g("NumAgprs"), g("TotalNumSgprs"), g("ScratchSize"), g("LDSByteSize"), occ, g("codeLenInByte")))
    print("# hipcc %s -S --cuda-device-only; scratch = bytes per lane the register allocator spilled (0 = none); occ = waves per SIMD the"
          % " ".join(x for x in kbuild.FLAGS if x not in ("-x", "hip", "-fPIC")))
    print("# compiler's register / LDS budget allows (launch bounds included); lds = static __shared__ bytes per workgroup")
    print("%-26s %-58s %5s %5s %5s %8s %7s %4s %7s" % ("file", "kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "occ", "code"))
    for r in rows:
        print("%-26s %-58s %5s %5s %5s %8s %7s %4s %7s" % (r[0], r[1][:58], *r[2:]))


if __name__ == "__main__":
    main()
