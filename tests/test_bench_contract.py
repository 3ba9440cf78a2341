"""bench.py prints one JSON line with the fields the driver reads (GPU: runs a reduced workload)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--groups", "6", "--cpu-groups", "2"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"]:
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "strong" and d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "traffic" in r and r["achieved"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c
    if "all_cores" in c:                       # present when the process may use more than one core
        assert c["all_cores"]["cores"] >= 2 and c["all_cores"]["value"] > 0
    assert d["sample_check"]["scored_equal"] and d["sample_check"]["entries_equal"]
    assert d["value"] > 0 and d["ms_per_step"] > 0
    e = d["e2e"]                               # cold end-to-end build of the same workload (never part of `value`)
    assert "error" not in e, e
    assert e["cold_s"] > 0 and e["gpu_part_s"] <= e["cold_s"] and e["file_bytes"] > 0 and e["scored"] == d["config"]["scored_per_step_per_gpu"]


@pytest.mark.gpu
def test_bench_weak_mode_and_group_output():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--groups", "4",
                          "--cpu-groups", "0", "--scaling", "weak", "--output", "group", "--e2e", "0"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["scaling"] == "weak" and d["value"] > 0 and "e2e" not in d


def _run_bench(argv, env=None, timeout=300):
    e = dict(os.environ)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, timeout=timeout, env=e)


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus N` with no launcher around it (the driver's command form): the parent starts N ranks as
    children, relays rank 0's line and their exit code.  CPU rehearsal of the launcher (--dry-run: gloo, no GPU call)."""
    out = _run_bench(["--gpus", "2", "--dry-run"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2


def test_bench_watchdog_ends_a_hung_job_with_an_error():
    """A rank that never reaches the collective: its peers' watchdog ends the job non-zero instead of waiting for ever."""
    import time
    t0 = time.time()
    out = _run_bench(["--gpus", "2", "--dry-run"], env={"IPK_BENCH_DRY_HANG": "1", "IPK_BENCH_WATCHDOG_S": "8"}, timeout=200)
    assert out.returncode != 0
    assert time.time() - t0 < 150
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_self_launched():
    """The N > 1 bench path end to end on a one-GPU box: both ranks pinned to GPU 0, gloo transport (RCCL refuses two ranks
    on one device), started by bench.py itself."""
    out = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--groups", "12"],
                     env={"IPK_BENCH_DEVICE": "0", "IPK_DIST_BACKEND": "gloo"}, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["exchange"] == "torch" and d["value"] > 0
    assert d["scaling"] == "strong" and "roofline" in d
    # the stages behind a step, timed after it: filter, shard files, rank 0's merge -- and the piece rule's arithmetic
    em = d["e2e_multi"]
    assert "error" not in em and em["kmers"] > 0 and em["entries"] > 0 and em["file_bytes"] > 16 * em["kmers"] + 8 * em["entries"]
    assert all(em[x] >= 0 for x in ("build_s", "filter_s", "shard_files_s", "merge_rank0_s"))
    assert d["pieces_used"] >= 1 and len(d["pieces_model"]["by_pieces"]) == 3
    # strong scaling: the two ranks together scored what one rank scores alone
    one = _run_bench(["--gpus", "1", "--steps", "1", "--warmup", "0", "--groups", "12", "--cpu-groups", "0", "--e2e", "0"], timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][0])
    assert d1["n_ranks_seen"] == 1 and d1["exchange"] == "none"
    assert round(d["value"] * d["ms_per_step"]) == round(d1["value"] * d1["ms_per_step"])
