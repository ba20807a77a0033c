"""The bench.py output contract, checked on the committed evidence (profiles/r02_g_bench.json is an unedited bench.py line
from an MI355X) and on the argument defaults -- no GPU needed."""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    return json.load(open(os.path.join(ROOT, "profiles", name)))


def test_committed_bench_line_has_every_contract_field():
    d = _line("r02_g_bench.json")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] in json.dumps(base) and d["n_gpus"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1000.0 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]      # whole-job hypotheses per second
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["unit"] == d["unit"]


def test_rocprof_summary_agrees_with_the_bench_line():
    """roofline.avg_launch_ms (HIP events inside bench.py) vs the committed rocprofv3 --stats average of the same kernel."""
    d = _line("r02_g_bench.json")
    kernel = d["roofline"]["kernel"]
    import csv
    avg_ns = None
    for row in csv.DictReader(open(os.path.join(ROOT, "profiles", "r02_kernel_stats.csv"))):
        if re.search(r"\b%s\(" % re.escape(kernel), row["Name"]):
            avg_ns = float(row["AverageNs"])
            break
    assert avg_ns is not None, kernel
    assert abs(avg_ns * 1e-6 - d["roofline"]["avg_launch_ms"]) / d["roofline"]["avg_launch_ms"] < 0.05


def test_bench_defaults_are_one_gpu_and_minutes():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1', src)
    steps = int(re.search(r'"--steps", type=int, default=(\d+)', src).group(1))
    assert 1 <= steps <= 100


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reported_world_is_the_launched_world_never_the_gpus_flag():
    """`--gpus 8` on one process must not print 8x the throughput of one GPU."""
    import pytest
    bench = _bench_module()

    class A:
        gpus = 1

    # plain single process
    assert bench.resolve_world(A, environ={}) == (1, 0, 0)
    # torchrun world that agrees with --gpus
    A.gpus = 4
    assert bench.resolve_world(A, environ={"WORLD_SIZE": "4", "RANK": "2", "LOCAL_RANK": "2"}) == (4, 2, 2)
    # torchrun world that disagrees: refuse (non-zero exit), both directions
    A.gpus = 8
    with pytest.raises(SystemExit) as e:
        bench.resolve_world(A, environ={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert e.value.code not in (0, None)
    A.gpus = 1
    with pytest.raises(SystemExit):
        bench.resolve_world(A, environ={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    # bare `python bench.py --gpus 8`: the ranks are spawned (before any GPU call), this process never measures
    A.gpus = 8
    spawned = []

    def fake_spawn(n):
        spawned.append(n)
        raise SystemExit(0)
    with pytest.raises(SystemExit):
        bench.resolve_world(A, environ={}, spawn=fake_spawn)
    assert spawned == [8]
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "a.gpus *" not in src and '"n_gpus": a.gpus' not in src           # numbers come from `world`
    assert "os._exit(0)" not in src                                           # the watchdog leaves non-zero
