"""bench.py's contract, on a small workload: the N = 1 line carries `roofline` (with traffic measured live by the
rocprofv3 child) and `cpu_baseline`; two ranks (gloo, both on device 0: the rehearsal of the N > 1 path one GPU
allows) run BASELINE configs[2]'s STRONG scaling -- the same reads cut into contiguous shards -- and report the
weak leg beside it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _line(out):
    return json.loads(out.strip().splitlines()[-1])


def test_single_gpu_line_has_roofline_traffic_and_cpu_baseline():
    r = subprocess.run([sys.executable, "bench.py", "--workload", "mini", "--steps", "3", "--warmup", "1", "--cpu-sample", "5000"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak" and d["unit"] == "reads/s"
    assert abs(d["value"] - d["config"]["global_reads_per_step"] * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1.05 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    # measured in this very run by the profiler child, and close to the algorithmic bytes (no wasted re-reads)
    assert roof["traffic_source"].startswith("live:"), roof["traffic_source"]
    # (this small genome's rows are 77 bytes at a 128-byte pitch: whole lines are fetched, so the bound is the pitch)
    import re
    row_bytes = int(re.search(r"rows x (\d+) B", d["config"]["workload"]).group(1))
    pitch = roof["index_bytes_in_hbm"] / (4 ** 9 + 1)
    assert 0.9 < roof["traffic"] / (roof["algorithmic_bytes_per_launch"] * pitch / row_bytes) < 1.2
    assert 0 < roof["infinity_cache_share"] <= 1 and roof["hbm_side_estimate_GBps"] <= roof["achieved"]
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["value"] > 0
    assert d["checks"]["gpu_equals_oracle_on_sample"] is True and d["checks"]["source_bucket_recovered"] > 0.97
    assert d["pruned"]["outputs_identical_to_headline_run"] is True
    assert d["pcie_inclusive"]["ms"] > 0


def test_two_ranks_strong_scaling_rehearsal():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", "bench.py", "--gpus", "2", "--backend", "gloo", "--device-override", "0", "--workload", "mini",
           "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--no-pmc"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong"
    assert d["config"]["global_reads_per_step"] == 100_000 and d["config"]["reads_per_gpu"] == 50_000
    assert abs(d["value"] - 100_000 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["weak_scaling"]["value"] > 0 and d["cpu_baseline"] is None
    assert d["pruned"]["outputs_identical_to_headline_run"] is True
    assert d["checks"]["source_bucket_recovered"] > 0.97
