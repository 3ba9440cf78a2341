#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a `hipcc -S --cuda-device-only` listing.

usage: isa_blocks.py LISTING.s SUBSTRING_OF_MANGLED_NAME [--dump OUT.s]
Prints, per label, the number of VALU / SALU / LDS / VMEM / SMEM instructions -- the static view that goes with the
SQ_INSTS_* counters of tools/sq_counters.py.
"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    text = open(path).read()
    m = None
    for mm in re.finditer(r"^(\S+):\s*; @\1\n", text, re.M):
        if key in mm.group(1):
            m = mm
            break
    if m is None:
        sys.exit("no kernel matching " + key)
    end = text.index(".Lfunc_end", m.end())
    body = text[m.end():end]
    if "--dump" in sys.argv:
        open(sys.argv[sys.argv.index("--dump") + 1], "w").write(m.group(0) + body)
    print(m.group(1))
    label, counts, order = "entry", {}, []
    tot = dict(V=0, S=0, DS=0, VM=0, SM=0)
    for line in body.splitlines():
        t = line.strip()
        if not t or t.startswith(";"):
            continue
        lm = re.match(r"^(\.LBB\S+):", t)
        if lm:
            label = lm.group(1)
            continue
        op = t.split()[0]
        if op.startswith("."):
            continue
        kind = ("DS" if op.startswith("ds_") else "VM" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else
                "SM" if op.startswith(("s_load", "s_buffer_load", "s_store")) else "V" if op.startswith("v_") else "S" if op.startswith("s_") else None)
        if kind is None:
            continue
        if label not in counts:
            counts[label] = dict(V=0, S=0, DS=0, VM=0, SM=0)
            order.append(label)
        counts[label][kind] += 1
        tot[kind] += 1
    for l in order:
        c = counts[l]
        print(f"{l:14s} V{c['V']:4d} S{c['S']:4d} DS{c['DS']:3d} VM{c['VM']:3d} SM{c['SM']:3d}")
    print("total", tot)
    for k in ("vgpr_count", "next_free_vgpr", "next_free_sgpr", "scratch", "private_segment_fixed_size"):
        for mm in re.finditer(r"^\s*\.amdhsa_" + k + r"\s+(\S+)", text[end:end + 6000], re.M):
            print(k, mm.group(1))
            break


if __name__ == "__main__":
    main()
