#!/usr/bin/env python3
"""FLAT memory instructions per kernel in a `hipcc -S --cuda-device-only` listing of ipk_amd/csrc/ipkgpu.hip.

A load or store through a pointer whose address space the compiler cannot see (one assembled from integers, or fetched from an
array of pointers) becomes flat_load / flat_store.  FLAT operations count on lgkmcnt as well as vmcnt, so the next wait for an LDS
read also waits for them -- in km_write_c_kernel that serialised eight memory round trips per wavefront.
usage: isa_flat.py [LISTING.s]   (without an argument the listing is made in a temporary directory; exit status 1 if any kernel
has FLAT instructions)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def listing():
    out = os.path.join(tempfile.mkdtemp(prefix="ipk_isa_"), "ipkgpu.s")
    cmd = [os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-S", "--cuda-device-only", "-w", os.path.join(ROOT, "ipk_amd", "csrc", "ipkgpu.hip"), "-o", out]
    subprocess.check_call(cmd)
    return out


def flat_by_kernel(path):
    text = open(path).read()
    res = {}
    for m in re.finditer(r"^(_Z\S+):\s*; @\1\n", text, re.M):
        end = text.index(".Lfunc_end", m.end())
        c = collections.Counter(l.split()[0] for l in text[m.end():end].splitlines() if l.startswith("\tflat_"))
        if c:
            res[m.group(1)] = dict(c)
    return res


if __name__ == "__main__":
    r = flat_by_kernel(sys.argv[1] if len(sys.argv) > 1 else listing())
    for k, v in r.items():
        print(k[:120], v)
    print("%d kernel(s) with FLAT memory instructions" % len(r))
    sys.exit(1 if r else 0)
