#!/bin/bash
# usage: tools/var_variant.sh "<bench args>" v1 v2 ...  -- bench.py per-kernel times with the scoring variant option set to each value
args=$1; shift
for v in "$@"; do
  timeout -k 10 300 python bench.py $args --variant $v --steps 5 --warmup 2 --e2e 0 --cpu-groups 0 > gpurun_out/vv_$v.json 2> gpurun_out/vv_$v.err
  python -c "
import json; j=json.load(open('gpurun_out/vv_$v.json')); print('variant $v', round(j['ms_per_step'],3), [(k['kernel'][:30],round(k['avg_launch_ms'],3)) for k in j['roofline'].get('kernels')])"
done
