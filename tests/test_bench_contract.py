"""The bench.py output contract, checked on the CODE that builds the line (bench.make_line on synthetic stage times) and on
the argument defaults -- no GPU needed. (The round's measured lines live in profiles/ as evidence, not as test input.)"""
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


STAGES = ["fps1", "ball1", "sa1", "p2", "fps2", "ball2", "sa2", "sa3", "fc"]


def test_make_line_has_every_contract_field_and_consistent_arithmetic():
    bench = _bench_module()
    stage_ms = [0.44, 0.43, 6.2, 0.26, 0.07, 0.05, 6.0, 0.8, 0.13]
    base = {"value": 104.0, "unit": "hyp/s", "cores": 16, "kind": "port", "sample": "192 of the 1000 hypotheses"}
    steps, world, elapsed = 20, 2, 0.31
    d = bench.make_line(STAGES, stage_ms, 0.035, elapsed, world, steps, 3, 1, 0, base, {"forward": {}})
    d = json.loads(json.dumps(d))                                    # it must survive the trip through JSON
    baseline = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] in json.dumps(baseline) and d["n_gpus"] == world and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["steps"] == steps and d["warmup"] == 3
    assert "workload" in d["config"] and "model" not in d["config"] and "configs[1]" in d["config"]["workload"]
    # whole-job hypotheses per second over ALL ranks; ms_per_step is one frame per rank
    assert abs(d["ms_per_step"] - 1e3 * elapsed / steps) < 1e-9
    assert abs(d["value"] - world * 1000.0 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["kernel"] == "sa1_kernel"      # the slowest MFMA stage
    assert r["peak"] == bench.PEAK_F32_MATRIX_TFLOPS
    assert abs(r["achieved"] - bench.SA1_FLOPS * 1000 / 6.2e-3 / 1e12) < 1e-9
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    assert abs(r["flops_per_launch"] - bench.SA1_FLOPS * 1000) < 1
    f = d["featurize"]
    assert f["bound"] == "hbm" and f["unit"] == "GB/s" and abs(f["frac"] - f["achieved"] / f["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["unit"] == d["unit"]
    assert set(d["stage_ms"]) == set(STAGES)
    # another stage dominating moves the roofline block with it
    stage_ms2 = list(stage_ms)
    stage_ms2[6] = 7.5
    assert bench.make_line(STAGES, stage_ms2, 0.035, elapsed, 1, steps, 3, 1, 0, None, None)["roofline"]["kernel"] == "sa2_kernel"


def test_dtoid_flop_model_is_consistent():
    """nominal = the survey's 39.7 + 45.96 n_t GFLOP; the executed count never exceeds it and the GEMM reassociation only
    switches on from 40 templates."""
    bench = _bench_module()
    for nt in (1, 10, 21, 39, 40, 160):
        nominal, executed = bench.dtoid_flops(nt)
        assert abs(nominal - (39.7e9 + 45.96e9 * nt)) < 1 and 0 < executed < nominal
    assert bench.dtoid_flops(160)[1] / 160 < bench.dtoid_flops(39)[1] / 39          # the GEMM form executes less per template


def test_a_failed_dtoid_leg_exits_non_zero_after_printing_the_line():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tail = src[src.index("    emit(dtoid_out)"):]
    assert "os._exit(4)" in tail and tail.index("emit(dtoid_out)") < tail.index("os._exit(4)")
    assert "GraphedForwardBackward" not in src                 # the graph-replay comparison lives in tools/bench_finetune.py


def test_bench_defaults_are_one_gpu_and_minutes():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert re.search(r'"--gpus", type=int, default=1', src)
    steps = int(re.search(r'"--steps", type=int, default=(\d+)', src).group(1))
    assert 1 <= steps <= 100


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
