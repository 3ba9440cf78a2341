#!/usr/bin/env python3
"""Sums rocprofv3 --pmc CSV output per kernel: {kernel: {counter: value per launch, "launches": n}}.

usage: sq_counters.py <out.json> <counter_collection.csv> [more csv ...]
Counters of several passes (rocprofv3 takes 8 SQ counters per pass) are merged per kernel name.
"""
import csv
import json
import re
import sys


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    acc = {}
    for path in paths:
        per = {}
        for r in csv.DictReader(open(path)):
            k = re.sub(r"\(.*", "", r["Kernel_Name"]).strip()
            d = per.setdefault(k, {})
            c = d.setdefault(r["Counter_Name"], {})
            c[r["Dispatch_Id"]] = c.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
            for extra in ("VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Workgroup_Size", "Grid_Size"):
                if extra in r and r[extra] not in ("", None):
                    d.setdefault("_" + extra, r[extra])
        for k, d in per.items():
            a = acc.setdefault(k, {})
            for c, v in d.items():
                if c.startswith("_"):
                    a[c[1:]] = v
                else:
                    a[c] = sum(v.values()) / len(v)
                    a["launches"] = len(v)
    json.dump(acc, open(out, "w"), indent=1, sort_keys=True)
    for k, a in acc.items():
        if "score" in k or "reduce" in k or "km_" in k:
            print(k, json.dumps(a, sort_keys=True))


if __name__ == "__main__":
    main()
