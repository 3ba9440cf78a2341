#!/bin/bash
# usage: tools/var_phase.sh "<bench args>" v1 v2 ...  -- ms per step and the key-major phase ("compact") of build variants (see tools/var_run.sh)
args=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset IPKGPU_LIB; else export IPKGPU_LIB=$PWD/ipk_amd/_variants/v_$v.so; fi
  timeout -k 10 300 python bench.py $args --steps 5 --warmup 2 --e2e 0 --cpu-groups 0 > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python -c "
import json; j=json.load(open('gpurun_out/var_$v.json')); p=j['phases_ms_per_step']; print('$v', round(j['ms_per_step'],3), 'compact', round(p['compact'],3), 'writer', round(p.get('km_write',0),3))"
done
