for f in 0 1 0 1; do
IPKGPU_DEBUG_FLAGS=$f python bench.py --e2e 0 --cpu-groups 0 --steps 5 --warmup 2 2>/dev/null | tail -1 > gpurun_out/ovf_f$f.json
python -c "
import json; d=json.load(open('gpurun_out/ovf_f$f.json')); p=d['phases_ms_per_step']; print($f, round(d['ms_per_step'],3), round(p['score'],3), round(p['score_main_kernel'],3), round(p['score_lds_reduce'],3), 'other in score', round(p['score']-p['score_main_kernel']-p['score_lds_reduce'],3))"
done
