#!/bin/bash
# usage (on the GPU box): tools/mkvar.sh ovfnoput -DIPK_OVF_NOPUT=1 (here, beforehand); then tools/ovf_probe.sh
# -- the big-list kernel of cfg2 (score_overflow_kernel) with and without its global atomics: rocprofv3 kernel stats of the in-tree
#    library and of the IPK_OVF_NOPUT build (results of the latter are wrong: timing only)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
unset IPKGPU_LIB
bash tools/prof_stats.sh ovf_with
export IPKGPU_LIB=$ROOT/ipk_amd/_variants/v_ovfnoput.so
bash tools/prof_stats.sh ovf_without
python3 - <<'PY'
import csv
for t in ("ovf_with", "ovf_without"):
    for r in csv.DictReader(open("gpurun_out/%s_kernel_stats.csv" % t)):
        if "score_overflow_kernel" in r["Name"]:
            print(t, r["Name"][:60], r["Calls"], "calls", round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
