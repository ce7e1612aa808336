"""bench.py's output contract (one JSON line on stdout with the driver's keys, the roofline and, at N = 1,
the CPU baseline), checked on a short run."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_prints_one_json_line_with_the_contract_keys(built):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["unit"] == "Mrays/s" and j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 1000.0                      # the target of BASELINE.json: >= 1 Gray/s on this config
    assert abs(j["value"] - 16777216 / (j["ms_per_step"] * 1e-3) / 1e6) < 1e-6 * j["value"]
    roof = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0.0 < roof["frac"] <= 1.0, "a roofline fraction is a fraction"
    assert roof["kernel"].startswith("trace_packet_rows_kernel<false, false, 2, ") and roof["algorithmic"]["node_rows"] > 0   # the instantiation, as rocprofv3 spells it
    assert roof["kernel_ms"] > 0 and "median" in roof["kernel_ms_is"] and roof["kernel_ms_mean"] > 0
    # committed counter passes are reported only for this very kernel on this very source tree; otherwise the line says why not
    assert (roof["traffic"] is not None) != ("traffic_rejected" in roof) or roof["traffic"] is None
    if roof["traffic"] is not None:
        assert roof["traffic_source"].count(j["source_sha16"]) == 1
    # the timed launches did the work: the last frame's digest equals the oracle's digest of the whole grid
    assert j["verified"]["hit_count"] == 16649551 and j["verified"]["rays"] == 16777216
    assert j["end_to_end_host_mrays"] > 100.0
    cpu = j["cpu_baseline"]
    assert cpu["kind"] in ("reference", "port") and cpu["cores"] >= 1 and cpu["value"] > 0 and cpu["unit"] == "Mrays/s"
    assert cpu["cores"] <= cpu["cpus_allowed"] and cpu["cpu_model"] and "median of 5" in cpu["sample"]   # one socket's cores, stated
