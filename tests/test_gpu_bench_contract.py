"""bench.py's one-line JSON contract (driver side): keys, types and the two extra objects, on a small workload."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_has_the_contract_fields():
    d = run_bench("--filters", "16384", "--steps", "5", "--warmup", "2", "--cpu-sample-filters", "256",
                  "--cpu-sample-seconds", "0.5")
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert key in d and isinstance(d[key], typ), key
    assert d["vs_baseline"] is None and d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] in ("weak", "strong") and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["achieved"] > 0
    assert "traffic" in r   # PMC figure for the default workload, None for others
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert d["value"] > 0 and abs(d["value"] - 16384 * 5 / (d["ms_per_step"] * 5e-3)) / d["value"] < 1e-6
    assert d["status_or"] == 0


def test_bench_other_workloads_run():
    d = run_bench("--workload", "orient", "--precision", "f32", "--filters", "8192", "--steps", "3", "--warmup", "1",
                  "--no-cpu-baseline")
    assert d["dtype"] == "f32" and "OrientationState" in d["metric"] and d["status_or"] == 0
    d = run_bench("--workload", "pose-mixed", "--filters", "8192", "--steps", "3", "--warmup", "1", "--no-cpu-baseline")
    assert d["value"] > 0
