# usage: tools/var_run.sh "<bench args>" v1 v2 ...   -- bench.py per-kernel times with ipk_amd/_variants/v_<name>.so ("base" = the in-tree library)
# (variants are built by hand: hipcc ... -D<KNOB>=<value> -o ipk_amd/_variants/v_<name>.so; knobs: IPK_QNW IPK_QTW IPK_QCAP IPK_ROUNDS IPK_WG_CHUNKS2 IPK_AACAP IPK_XPNW)
args=$1; shift
for v in "$@"; do
  if [ $v = base ]; then unset IPKGPU_LIB; else export IPKGPU_LIB=$PWD/ipk_amd/_variants/v_$v.so; fi
  timeout -k 10 300 python bench.py $args --steps 5 --warmup 2 --e2e 0 --cpu-groups 0 > gpurun_out/var_$v.json 2> gpurun_out/var_$v.err
  python -c "
import json; j=json.load(open('gpurun_out/var_$v.json')); print('$v', round(j['ms_per_step'],3), [(k['kernel'],round(k['avg_launch_ms'],3)) for k in j['roofline'].get('kernels')])"
done
