"""DNA k = 13, 14 at a cfg3-like shape (the share of one of 8 ranks: 125 groups x 2 matrices x 10 000 sites, alpha 0.05): time per pass of the
key-major build, keys and entries; k = 12 beside it.  Usage: python tools/k13_probe.py [groups] [sites]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import ipk_amd
from ipk_amd.synth import synth_matrices
from ipk_amd import distributed as D

groups = int(sys.argv[1]) if len(sys.argv) > 1 else 125
sites = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
mats = synth_matrices(2 * groups, sites, 4, 0.05, 42)
mg = np.repeat(np.arange(groups, dtype=np.uint32), 2)
dev = torch.from_numpy(mats).cuda()
eng = ipk_amd.Engine(0)
if os.environ.get("IPKGPU_DEBUG_FLAGS"):
    eng.set_option("debug_flags", int(os.environ["IPKGPU_DEBUG_FLAGS"]))
from ipk_amd import engine as E
ks = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [12, 13, 14]
for k in ks:
    eps = ipk_amd.log_threshold(1.5, 4, k)
    ts = []
    for it in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        db, parts = D.build_db_shard(eng, dev, mg, k, eps, 4)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        nk, ne = db.num_keys, db.num_entries
        print(f"  pass {it}: {ts[-1]:.1f} ms wall; device total {parts.time_ms(E.T_TOTAL):.1f}, scoring {parts.time_ms(E.T_SCORE):.1f} (count {parts.time_ms(E.T_XP_COUNT):.1f}, "
              f"batches {parts.time_ms(E.T_SCORE_LAUNCHES):.0f}, write {parts.time_ms(E.T_XP_WRITE):.1f}, reduce {parts.time_ms(E.T_SCORE_REDUCE):.1f}), key-major {parts.time_ms(E.T_COMPACT):.1f} (writer {parts.time_ms(E.T_KM_WRITE):.1f}), db {db.time_ms():.1f}", flush=True)
        if it == 3:
            # filter values over all keys (a wavefront per key, launched in spans of 2^24 keys): sampled keys of every span against the oracle's formula
            from oracle import ipk_oracle as co          # (a diagnostic tool, not the product: the oracle only checks)
            from ipk_amd.engine import _device_tensor
            thr = ipk_amd.score_threshold(1.5, 4, k)
            N = 2 * groups - 1
            db.filter_mif0(eng, N, thr)
            fv64 = db.filter_values(f64=True)
            off = db.key_offsets().astype(np.int64)
            ent = _device_tensor(db.entries_device_ptr(), (ne, 2), "int32", db)
            worst = 0.0
            for i in np.linspace(0, nk - 1, 600).astype(np.int64).tolist():
                sc = ent[off[i]:off[i + 1], 1].cpu().numpy().view(np.float32)
                ref = co.mif0(sc, N, thr)
                worst = max(worst, abs(fv64[i] - ref) / max(1.0, abs(ref)))
            assert worst <= 1e-9, worst
            print(f"  filter values of 600 keys over {nk} keys agree with the oracle (worst relative difference {worst:.1e}), {db.filter_time_ms():.2f} ms", flush=True)
        db.free(); parts.free()
    st = eng.last_stats() if hasattr(eng, "last_stats") else None
    print(f"k={k} groups={groups} sites={sites}: {min(ts[1:]):.2f} ms per pass (first {ts[0]:.1f}), keys {nk}, entries {ne}", flush=True)
