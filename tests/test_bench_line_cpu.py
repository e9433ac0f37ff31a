"""CPU: the contract of bench.py's one JSON line, checked on the newest committed line (profiles/r*_bench.json), the
roofline arithmetic bench.py applies, and the plumbing of its own multi-rank launcher (no GPU needed)."""
import json
import os
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _newest(suffix):
    names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith(suffix))
    return os.path.join(ROOT, "profiles", names[-1])


def _line():
    with open(_newest("_bench.json")) as f:
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
    # the matrix roof is the headline (the roof the measurement plan names for the conv kernels); the memory side of the
    # same launch is reported beside it (lines of earlier rounds had it the other way round)
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
    with open(_newest("_pmc_traffic.json")) as f:
        kernels = json.load(f)["kernels"]
    chain = [v for k, v in kernels.items() if "conv3x3_chain" in k]
    assert len(chain) == 1
    assert chain[0]["traffic_bytes_per_launch"] == chain[0]["read_bytes_per_launch"] + chain[0]["write_bytes_per_launch"]
    assert d["roofline"]["traffic"] in (None, chain[0]["traffic_bytes_per_launch"]) or \
        abs(d["roofline"]["traffic"] - chain[0]["traffic_bytes_per_launch"]) < 0.02 * chain[0]["traffic_bytes_per_launch"]


def test_bench_quotes_the_newest_committed_traffic_summary():
    import bench
    assert bench.pmc_traffic_file() == _newest("_pmc_traffic.json")


def test_plain_multi_gpu_start_launches_one_fresh_process_per_rank(tmp_path, capfd):
    """`python bench.py --gpus N` without a launcher: N child processes with the rendezvous environment of
    torch.distributed.run, rank 0's JSON line relayed, the worst exit status returned; the parent touches no GPU."""
    import bench
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        env = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
        open(os.path.join(os.path.dirname(__file__), "rank%s.json" % env["RANK"]), "w").write(json.dumps([env, sys.argv[1:]]))
        print("noise before the line")
        print(json.dumps({"rank": env["RANK"], "n_gpus": int(env["WORLD_SIZE"])}))
        sys.exit(3 if env["RANK"] == "2" and "--fail" in sys.argv else 0)
    """))
    rc = bench.launch_ranks(3, ["--gpus", "3", "--steps", "2"], script=str(script))
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0 and len(out) == 1 and json.loads(out[0]) == {"rank": "0", "n_gpus": 3}
    seen = [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(3)]
    assert [e["RANK"] for e, _ in seen] == ["0", "1", "2"] and [e["LOCAL_RANK"] for e, _ in seen] == ["0", "1", "2"]
    assert all(e["WORLD_SIZE"] == "3" and e["MASTER_ADDR"] == "127.0.0.1" for e, _ in seen)
    assert len({e["MASTER_PORT"] for e, _ in seen}) == 1 and int(seen[0][0]["MASTER_PORT"]) > 0
    assert all(argv == ["--gpus", "3", "--steps", "2"] for _, argv in seen)
    assert bench.launch_ranks(3, ["--fail"], script=str(script)) == 3
