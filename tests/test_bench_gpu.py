"""bench.py's contract, on a small workload: the N = 1 line carries `roofline` (with traffic measured live by the
rocprofv3 child), `cpu_baseline` and the `skewed` leg (genome-like genome); several ranks (gloo, all on device 0: the
rehearsal of the N > 1 path one GPU allows -- two ranks, and four: the GPU box lets six processes on its card at once,
and the test runner itself is one of them) run BASELINE configs[2]'s STRONG scaling -- the same reads cut into contiguous shards -- and report the weak leg
and the per-rank times beside it."""
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
    r = subprocess.run([sys.executable, "bench.py", "--workload", "mini", "--steps", "3", "--warmup", "1", "--cpu-sample", "5000", "--locator-cpu-seconds", "2"],
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
    # the same geometry on the genome-like genome, beside the headline: parity on its own sample, both kernels' times,
    # the share of rows under the distinguishability threshold and of reads with a candidate
    sk = d["skewed"]
    assert sk["checks"]["gpu_equals_oracle_on_sample"] is True and sk["checks"]["parity_sample_reads"] > 0
    assert sk["pruned"]["outputs_identical_to_default_kernel"] is True and sk["pruned"]["ms_per_step"] > 0
    assert 0.5 < sk["rows_passing_distinguishability"] < 1.0 and 0.8 < sk["checks"]["reads_with_candidates"] <= 1.0
    assert 0 < sk["roofline"]["frac"] < 1.05
    assert "roofline_large_index" not in d and "verifier" not in d     # (the 4.6 GB and verifier legs belong to the default `egu` run)
    # north_star's second subsystem, the locator's candidate scan, on the headline's and on the skewed leg's candidates:
    # kernel times, occurrences, a fraction of the HBM peak, the CPU oracle beside it and parity on its sample
    for loc in (d["locator"], sk["locator"]):
        assert "error" not in loc, loc
        assert loc["candidates"] > 0 and loc["ms_scan"] > 0 and loc["ms_replay"] > 0 and loc["occurrences"] > 0
        assert loc["roofline"]["bound"] == "hbm" and 0 < loc["roofline"]["frac"] < 1
        assert abs(loc["roofline"]["frac"] - loc["roofline"]["achieved"] / loc["roofline"]["peak"]) < 1e-9
        assert loc["cpu_baseline"]["kind"] == "port" and loc["cpu_baseline"]["cores"] == 1 and loc["cpu_baseline"]["value"] > 0
        assert loc["checks"]["gpu_equals_oracle_on_sample"] is True and loc["checks"]["parity_sample_candidates"] > 0
    assert d["locator"]["checks"]["true_candidates_located_at_the_simulated_offset"] > 0.97
    assert sk["locator"]["occurrences"] > d["locator"]["occurrences"]       # repeats: the genome-like genome's signature


@pytest.mark.parametrize("ranks", [2, 4])
def test_strong_scaling_rehearsal(ranks):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(29533 + ranks), "bench.py", "--gpus", str(ranks), "--backend", "gloo", "--device-override", "0",
           "--workload", "mini", "--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--no-pmc"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == ranks and d["scaling"] == "strong"
    per = [100_000 * (i + 1) // ranks - 100_000 * i // ranks for i in range(ranks)]
    assert d["config"]["global_reads_per_step"] == 100_000 and d["config"]["reads_per_gpu"] == per[0]
    assert abs(d["value"] - 100_000 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["weak_scaling"]["value"] > 0 and d["cpu_baseline"] is None
    assert d["pruned"]["outputs_identical_to_headline_run"] is True
    assert d["checks"]["source_bucket_recovered"] > 0.97
    assert "skewed" not in d and "roofline_large_index" not in d and "verifier" not in d           # N = 1 legs
    # every rank's view of the timed region: its shard, its step time, its kernels' time, its setup
    pr = d["per_rank"]
    assert pr["reads"] == per and len(pr["step_ms"]) == ranks
    assert abs(max(pr["step_ms"]) - d["ms_per_step"]) < 0.002               # the headline is the slowest rank (3 decimals kept)
    assert all(0 < k <= s + 0.002 for k, s in zip(pr["kernel_ms_mean"], pr["step_ms"]))
    assert all(a <= b for a, b in zip(pr["kernel_ms_min"], pr["kernel_ms_max"])) and all(s > 0 for s in pr["setup_wall_s"])
