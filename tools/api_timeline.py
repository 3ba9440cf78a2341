"""Host-side view of the last build step of a rocprofv3 --hip-trace --kernel-trace run: every HIP API call between two prefix kernels
with its duration, plus the kernels, on one time axis.  usage: python tools/api_timeline.py <dir> [min_us=2]"""
import csv, glob, re, sys
d = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
ker, api = [], []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("ipkgpu::", "")[:60]))
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A " + r["Function"]))
ker.sort()
st = [k[0] for k in ker if "prefix_max" in k[2]]
lo, hi = st[-2], st[-1]
# from the end of the last but one step's writer to the start of the last step's scoring kernel
ops = sorted([o for o in ker + api if lo - 400000 <= o[0] < hi])
t0 = lo
tot = {}
for s, e, n in ops:
    if n.startswith("A "):
        tot[n] = tot.get(n, 0) + (e - s)
    if (e - s) / 1e3 >= min_us or n.startswith("K "):
        print("%9.1f  %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
print("---- API time by function (us) over the window")
for n, v in sorted(tot.items(), key=lambda x: -x[1])[:15]:
    print("%8.1f  %s" % (v / 1e3, n))
