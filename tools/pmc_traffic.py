#!/usr/bin/env python3
"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into a per-kernel HBM traffic summary (JSON).

Per /opt/skills/guides/MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KiB, collected in separate passes (TCC
slots); on gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is
taken as is (exact for 16-B/lane stores; the 8-B/lane pair stores of the scoring kernel are uncalibrated -- stated in the
JSON).

usage: pmc_traffic.py <fetch_csv> <write_csv> <workload> <out.json> <kernel-substring> [more substrings ...]
(a substring may be several parts joined by '&': all must occur in the kernel name -- template instantiations;
'name=substring' stores the kernel under `name`).  <workload> must be the bench config's name (cfg2, cfg4 ...) for bench.py to use it.
The first substring's kernel is also written at top level ("kernel", "hbm_bytes_per_launch"): bench.py reads that.
"""
import csv
import json
import sys


def per_launch(path, counter, kernel_sub):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and all(part in r["Kernel_Name"] for part in kernel_sub.split("&")):
            vals.setdefault(r["Dispatch_Id"], 0.0)
            vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    if not vals:
        return None, 0
    v = list(vals.values())
    return sum(v) / len(v), len(v)


def main():
    fetch_csv, write_csv, workload, out = sys.argv[1:5]
    subs = sys.argv[5:]
    rec = {"workload": workload,
           "corrections": "FETCH_SIZE x2 (gfx950 reports half of wide coalesced reads); WRITE_SIZE as is (8-B/lane stores uncalibrated); KiB -> bytes",
           "kernels": {}}
    for i, sub in enumerate(subs):
        name, _, pat = sub.rpartition("=")                 # "name=pattern": the key under which bench.py looks the kernel up
        sub, pat = (name or pat), pat
        f, nf = per_launch(fetch_csv, "FETCH_SIZE", pat)
        w, nw = per_launch(write_csv, "WRITE_SIZE", pat)
        if f is None or w is None:
            continue
        k = {"FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w, "launches": [nf, nw],
             "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
        rec["kernels"][sub] = k
        if i == 0:
            rec.update(kernel=sub, **k)
    json.dump(rec, open(out, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
