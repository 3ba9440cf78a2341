# usage: tools/var_run.sh <config> v1 v2 ...   -- bench.py per-kernel times with ipk_amd/_variants/v_<name>.so ("base" = the in-tree library)
cfg=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset IPKGPU_LIB; else export IPKGPU_LIB=$PWD/ipk_amd/_variants/v_$v.so; fi
  timeout -k 10 200 python bench.py --config $cfg --steps 3 --warmup 1 --e2e 0 --cpu-groups 0 > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python -c "
import json; j=json.load(open('gpurun_out/var_$v.json')); print('$v', round(j['ms_per_step'],3), [(k['kernel'],round(k['avg_launch_ms'],3)) for k in j['roofline'].get('kernels')], j.get('timers'))"
done
