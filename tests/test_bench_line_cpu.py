"""CPU: the contract of bench.py's one JSON line, checked on the committed line of the final tree
(profiles/r02l_bench.json) and on the roofline arithmetic bench.py applies (no GPU needed)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    with open(os.path.join(ROOT, "profiles", "r02l_bench.json")) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def test_committed_bench_line_has_the_contract_fields():
    d = _line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["metric"].startswith("RRDBNet x4 HR Mpix/s") and d["unit"] == "HR Mpix/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = pixels of the batch / time of a step
    assert abs(d["value"] - 16 * 512 * 512 / (d["ms_per_step"] * 1e-3) / 1e6) < 0.01 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0.0 < r["frac"] < 1.0
    assert r["peak"] == (8000.0 if r["bound"] == "hbm" else 2500.0)
    # the roof that binds is the memory one exactly when the launch sits under the ridge of the two peaks
    assert (r["flop_per_byte"] < r["ridge_flop_per_byte"]) == (r["bound"] == "hbm")
    other = r["mfma_side"] if r["bound"] == "hbm" else r["hbm_side"]
    assert 0.0 < other["frac"] < 1.0
    # achieved = algorithmic bytes (or flops) per launch / the launch's measured duration
    per_launch = r["algorithmic_bytes_per_launch"] if r["bound"] == "hbm" else r["flop_per_launch"]
    scale = 1e9 if r["bound"] == "hbm" else 1e12
    assert abs(per_launch / (r["avg_launch_us"] * 1e-6) / scale - r["achieved"]) < 0.01 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.95 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


def test_committed_traffic_summary_is_what_the_line_quotes():
    d = _line()
    with open(os.path.join(ROOT, "profiles", "r02l_pmc_traffic.json")) as f:
        kernels = json.load(f)["kernels"]
    chain = [v for k, v in kernels.items() if "conv3x3_chain" in k]
    assert len(chain) == 1
    assert chain[0]["traffic_bytes_per_launch"] == chain[0]["read_bytes_per_launch"] + chain[0]["write_bytes_per_launch"]
    assert d["roofline"]["traffic"] in (None, chain[0]["traffic_bytes_per_launch"]) or \
        abs(d["roofline"]["traffic"] - chain[0]["traffic_bytes_per_launch"]) < 0.02 * chain[0]["traffic_bytes_per_launch"]
