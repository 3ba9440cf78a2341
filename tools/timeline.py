"""Timeline of the last build step in a rocprofv3 kernel trace (+ memory copies): start, duration and the idle gap before each
operation.  usage: python tools/timeline.py <dir with *_kernel_trace.csv> [marker kernel substring = prefix_max]"""
import csv, glob, re, sys
d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "prefix_max"
ops = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ops.sort()
starts = [i for i, o in enumerate(ops) if marker in o[2]]
if len(starts) < 2:
    sys.exit("no two steps found")
lo, hi = starts[-2], starts[-1]
t0 = ops[lo][0]
prev_end = t0
busy = 0
for s, e, n in ops[lo:hi]:
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"^void ", "", n).replace("ipkgpu::", "")
    n = re.sub(r"rocprim::.*scan_impl.*", "rocprim scan", n)
    n = re.sub(r"rocprim::.*init_lookback.*", "rocprim scan init", n)
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:110]))
    busy += e - s
    prev_end = max(prev_end, e)
print("step span %.1f us, busy %.1f us, idle %.1f us" % ((ops[hi][0] - t0) / 1e3, busy / 1e3, (ops[hi][0] - t0 - busy) / 1e3))
