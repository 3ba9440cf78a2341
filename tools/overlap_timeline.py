#!/usr/bin/env python3
"""usage: overlap_timeline.py <out.json> <seq trace dir> <par trace dir>
From two rocprofv3 --kernel-trace runs of tools/overlap_probe.py (--phase seq / --phase par): per kernel the mean duration when the
two half-workloads run one after the other and when they run from two threads, and, for the concurrent run, how long a scoring
kernel and a reduce / writer kernel of the OTHER context were actually on the device together."""
import csv
import glob
import json
import re
import sys


def load(d):
    ops = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"])
            n = re.sub(r"^void ", "", n).replace("ipkgpu::", "")
            n = re.sub(r"<.*", "", n)
            ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "")))
    ops.sort()
    t0, t1 = ops[0][0], ops[-1][1]
    return [o for o in ops if o[0] >= t0 + 0.5 * (t1 - t0)]          # the timed half (the first half is warm-up)


def summary(ops):
    per = {}
    for s, e, n, q in ops:
        a = per.setdefault(n, [0, 0.0])
        a[0] += 1; a[1] += (e - s) / 1e3
    return {n: {"launches": c, "mean_us": t / c} for n, (c, t) in per.items() if t / c > 50}


def overlap(ops, a_sub, b_subs):
    """time (us) during which a kernel matching a_sub and a kernel of another queue matching one of b_subs ran together"""
    A = [(s, e, q) for s, e, n, q in ops if a_sub in n]
    B = [(s, e, q) for s, e, n, q in ops if any(x in n for x in b_subs)]
    tot = 0
    for s, e, q in A:
        for s2, e2, q2 in B:
            if q2 != q and s2 < e and e2 > s:
                tot += min(e, e2) - max(s, s2)
    return tot / 1e3, sum(e - s for s, e, _ in A) / 1e3, sum(e - s for s, e, _ in B) / 1e3


def main():
    out, dseq, dpar = sys.argv[1:4]
    seq, par = load(dseq), load(dpar)
    ov, ta, tb = overlap(par, "score_quad_kernel", ("reduce_buckets", "km_write", "km_count"))
    span = lambda ops: (ops[-1][1] - ops[0][0]) / 1e3
    rec = {"what": "two half-workloads of cfg2 on two contexts (two HIP streams) of one GPU: one after the other vs from two threads",
           "sequential": {"span_us": span(seq), "kernels": summary(seq)},
           "concurrent": {"span_us": span(par), "kernels": summary(par),
                          "scoring_us": ta, "reduce_and_writer_us": tb, "scoring_with_other_contexts_reduce_or_writer_us": ov,
                          "share_of_reduce_and_writer_time_under_a_scoring_kernel": ov / tb if tb else None}}
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(rec["concurrent"], sort_keys=True)[:1500])


if __name__ == "__main__":
    main()
