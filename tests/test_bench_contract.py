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
