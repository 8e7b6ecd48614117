"""bench.py end to end WITHOUT a GPU: the script itself (argument handling, the analysis, the timed loop, the roofline
and cpu_baseline objects, the solve legs, the JSON line) run against the CPU build of the engine with torch.cuda
stubbed out.  Nothing about speed is learnt here -- only that the driver's command cannot fail on a script error,
and that the line has the fields the contract names."""
import json
import os
import sys

import pytest

from conftest import ROOT

CPUEMU = os.path.join(ROOT, "tests", "helpers", "libmi355_cpuemu.so")


def run_bench(monkeypatch, capsys, argv):
    import torch
    from suitesparse_amd import engine, symbolic
    if not os.path.exists(CPUEMU):
        pytest.skip("libmi355_cpuemu.so not built")
    monkeypatch.setattr(engine, "LIB_PATH", CPUEMU)
    monkeypatch.setattr(engine, "_LIB", None)
    symbolic._ready = False
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: None)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a: None)
    real_ones = torch.ones
    monkeypatch.setattr(torch, "ones", lambda *a, **k: real_ones(*a, **{**k, "device": "cpu"}))
    monkeypatch.setattr(engine, "mfma_f64_rate", lambda d=0: 0.0)        # a timing probe of the real device
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    sys.path.insert(0, ROOT)
    import importlib
    import bench
    importlib.reload(bench)
    try:
        bench.main()
    finally:
        symbolic._ready = False
    out = capsys.readouterr().out.strip().splitlines()
    assert len(out) == 1, out
    return json.loads(out[0])


@pytest.mark.parametrize("problem,grid,extra", [("lap3d", 7, []), ("helm", 5, []), ("lap2d", 24, ["--ordering", "metis"]),
                                                ("elast", 3, ["--no-cpu"])])
def test_bench_line_has_the_contract_fields(monkeypatch, capsys, problem, grid, extra):
    d = run_bench(monkeypatch, capsys, ["--problem", problem, "--grid", str(grid), "--cpu-grid", str(grid), "--steps", "2",
                                        "--warmup", "1"] + extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["unit"] == "GFLOP/s"
    assert d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None and d["value"] >= 0 and d["ms_per_step"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["peak"] == 78.6 and r["unit"] == "TFLOP/s" and "traffic" in r
    if "--no-cpu" not in extra:
        c = d["cpu_baseline"]
        assert c["kind"] == "reference" and c["value"] > 0 and c["cores"] >= 1 and c["unit"] == "GFLOP/s"
        assert "omp_get_max_threads" in c["threads"] and c["L_ordering"] == "given"
        assert isinstance(d["config"]["solve"], list) and [s["nrhs"] for s in d["config"]["solve"]] == [1, 64]
    if "metis" in extra:
        assert "METIS" in d["config"]["workload"] and "METIS" in d["cpu_baseline"]["sample"]
