import sys, time, numpy as np
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import ipk_amd
from ipk_amd.synth import synth_matrices, CONFIGS
cfg = CONFIGS["cfg2"]
n = 2000
mats = np.concatenate([synth_matrices(250, 10000, 4, 0.05, 42, first_mat=i) for i in range(0, n, 250)])
groups = np.repeat(np.arange(n // 2, dtype=np.uint32), 2)
eps = ipk_amd.log_threshold(1.5, 4, 10)
eng = ipk_amd.Engine(0)
for i in range(3):
    t = time.perf_counter(); r = eng.score_groups(mats, groups, 10, eps); dt = time.perf_counter() - t
    print("host-buffer call (pageable H2D + group-major CSR): %.1f ms, device part %.1f ms, emitted %d" % (dt * 1e3, r.time_ms(0), r.emitted))
    r.free()
