# usage: tools/var_flags.sh <config> <groups> f1 f2 ...   -- per-kernel times with IPKGPU_DEBUG_FLAGS=f
cfg=$1; g=$2; shift 2
for f in "$@"; do
  IPKGPU_DEBUG_FLAGS=$f timeout -k 10 200 python bench.py --config $cfg --groups $g --steps 3 --warmup 1 --e2e 0 --cpu-groups 0 > gpurun_out/vf_$f.json 2> gpurun_out/vf_$f.err
  python -c "
import json; j=json.load(open('gpurun_out/vf_$f.json')); print($f, round(j['ms_per_step'],3), [(k['kernel'],round(k['avg_launch_ms'],3)) for k in j['roofline'].get('kernels')])"
done
