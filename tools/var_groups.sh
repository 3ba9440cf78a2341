# usage: tools/var_groups.sh <config> g1 g2 ...   -- per-kernel times of bench.py at several group counts
cfg=$1; shift
for g in "$@"; do
  timeout -k 10 200 python bench.py --config $cfg --groups $g --steps 3 --warmup 1 --e2e 0 --cpu-groups 0 > gpurun_out/vg_$g.json 2> gpurun_out/vg_$g.err
  python -c "
import json; j=json.load(open('gpurun_out/vg_$g.json')); print($g, round(j['ms_per_step'],3), [(k['kernel'],round(k['avg_launch_ms'],3)) for k in j['roofline'].get('kernels')])"
done
