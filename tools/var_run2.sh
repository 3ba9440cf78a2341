# usage: tools/var_run2.sh "<bench args>" v1 v2 ...   -- like var_run.sh with free bench arguments
args=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset IPKGPU_LIB; else export IPKGPU_LIB=$PWD/ipk_amd/_variants/v_$v.so; fi
  timeout -k 10 200 python bench.py $args --steps 10 --warmup 3 --e2e 0 --cpu-groups 0 > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python -c "
import json; j=json.load(open('gpurun_out/var_$v.json')); print('$v', round(j['ms_per_step'],3), [(k['kernel'],round(k['avg_launch_ms'],3)) for k in j['roofline'].get('kernels')])"
done
