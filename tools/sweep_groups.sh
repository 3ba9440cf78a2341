#!/bin/bash
# per-call cost of the hot path at small group counts (one GPU): what a rank of a strong-scaled run sees
for g in "$@"; do
  timeout -k 10 200 python bench.py --groups $g --steps 20 --warmup 5 --cpu-groups 0 --e2e 0 2>/dev/null > /tmp/sweep_$g.json
  python - <<PY
import json
j = json.load(open("/tmp/sweep_$g.json"))
p = j["phases_ms_per_step"]
print("groups %4d  ms/step %7.3f  G/s %6.1f  main %.3f reduce %.3f compact %.3f prefix %.3f device_total %.3f" % ($g, j["ms_per_step"], j["value"] / 1e9, p["score_main_kernel"], p["score_lds_reduce"], p["compact"], p["prefix"], p["device_total"]))
PY
done
