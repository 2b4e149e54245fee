"""Host-side pieces of bench.py that run without a GPU: the ordering of the CPU legs and the per-pass trace."""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def test_cpu_legs_are_deferred_until_the_gpu_legs_are_done(monkeypatch):
    bench = importlib.import_module("bench")
    ran = []
    monkeypatch.setattr(bench, "_DEFERRED", None)
    bench._later(lambda: ran.append("now"))
    assert ran == ["now"], "a single-workload run executes its CPU leg in place"
    monkeypatch.setattr(bench, "_DEFERRED", [])
    bench._later(lambda: ran.append("a"))
    bench._later(lambda: ran.append("b"))
    assert ran == ["now"] and len(bench._DEFERRED) == 2, "a run over all workloads collects them"
    for fn in bench._DEFERRED:
        fn()
    assert ran == ["now", "a", "b"], "and runs them in order afterwards"


def test_pass_trace_is_off_unless_asked_for(monkeypatch, capsys):
    bench = importlib.import_module("bench")
    monkeypatch.delenv("MDG_BENCH_TRACE", raising=False)
    tr = bench._PassTrace("x")
    tr.start(); tr.tick(); tr.done()
    assert tr.t == [] and capsys.readouterr().err == ""
    monkeypatch.setenv("MDG_BENCH_TRACE", "1")
    tr = bench._PassTrace("leg")
    tr.start(); tr.tick(); tr.tick(); tr.done()
    err = capsys.readouterr().err
    assert len(tr.t) == 2 and "[trace leg] ms per pass" in err and "collections" in err


def _fake_secondary():
    return {"metric": "MD steps/sec (fwd+adjoint), 4096-bead SchNet CG water NHC", "value": 3019.123456789, "unit": "MD steps/s",
            "n_gpus": 1, "steps": 28, "warmup": 6, "ms_per_step": 26.5, "dtype": "bf16 filter MFMA operands, f32 accumulate",
            "config": {"replicas_per_gpu": 8, "dist": {"per_rank_ms": [26.5]}, "single_system": {"md_steps_per_s": 1233.0,
                                                                                                 "us_per_md_step": 811.0}},
            "roofline": {"bound": "mfma", "kernel": "k" * 400, "achieved": 117.0, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.047,
                         "traffic": 1.0e8, "kernel_ms": 0.176, "note": "n" * 2000, "step_roof": {"frac": 0.19, "model": "m" * 500}},
            "cpu_baseline": {"value": 0.1, "unit": "MD steps/s", "cores": 16, "kind": "port", "sample": "s" * 600,
                             "parity_sampled": {"max_abs_dq": 1e-6, "max_abs_dg": 2e-5, "rel_dtheta": 4e-4, "note": "x" * 900}}}


def test_secondary_workloads_become_flat_scalars_and_short_lines():
    """the driver keeps the scalar keys of `config` and the tail of stdout -- every secondary workload must
    survive both: flat scalar keys on the headline, one compact JSON line each before it."""
    import json
    bench = importlib.import_module("bench")
    rec = _fake_secondary()
    flat = bench._flat("schnet4096", rec)
    assert flat["schnet4096_md_steps_per_s"] == 3019.12 and flat["schnet4096_ms_per_pass"] == 26.5
    assert flat["schnet4096_step_roof_frac"] == 0.19 and flat["schnet4096_kernel_frac"] == 0.047
    assert flat["schnet4096_cpu_steps_per_s"] == 0.1 and flat["schnet4096_parity_max_abs_dq"] == 1e-6
    assert all(isinstance(v, (int, float, str)) for v in flat.values()), "scalars only: nested values are dropped by the driver"
    line = bench._line("schnet4096", rec)
    assert len(line) < 1500 and "\n" not in line
    back = json.loads(line)
    assert back["workload"] == "schnet4096" and back["roofline"]["step_roof_frac"] == 0.19 and back["parity"]["rel_dtheta"] == 4e-4
    assert "error" in json.loads(bench._line("lj4096", {"error": "RuntimeError: x"}))
    assert bench._flat("lj4096", {"error": "boom"}) == {"lj4096_error": "boom"}


def test_compact_headline_drops_notes_and_keeps_every_contract_key():
    bench = importlib.import_module("bench")
    head = {"metric": "m", "value": 1.23456789e7, "unit": "u", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 22.3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 300, "dist": {"backend": "nccl", "per_rank_ms": [1.0]}},
            "roofline": {"bound": "valu", "achieved": 26.1, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.166, "traffic": None,
                         "note": "n" * 3000},
            "cpu_baseline": {"value": 124.0, "unit": "MD steps/s", "cores": 16, "kind": "port", "sample": "s" * 700}}
    c = bench._compact(head)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c
    assert "note" not in c["roofline"] and len(c["cpu_baseline"]["sample"]) <= 150 and c["value"] == 1.23456789e7
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(c["roofline"])


def test_headline_config_puts_every_baseline_config_first_and_keeps_the_workload_short():
    """the driver's parser keeps the first ~20 scalar keys of `config` and cuts strings at 100
    characters -- the rate, time per pass and dominant-kernel fraction of configs #3 / #4 / #5 and the one-system-per-GPU leg
    come right after the workload string, parity / dtype scalars after them."""
    bench = importlib.import_module("bench")
    cfg = {"workload": "w" * 140, "replicas_per_gpu": 16384, "md_steps_per_pass": 1, "parallelism": "replica-dp1", "loss": 0.9,
           "dist": {"backend": "nccl"}, "rdf_fused_into_trajectory_kernels": True, "rdf_fused": "fine-grid histogram",
           "exvol108_md_steps_per_s": 3.4e7, "exvol108_dtype": "f32"}
    for name in ("schnet4096", "lj4096", "water192"):
        cfg.update({name + "_md_steps_per_s": 1.0, name + "_ms_per_pass": 2.0, name + "_dtype": "f32", name + "_kernel_frac": 0.1,
                    name + "_parity_max_abs_dq": 1e-6})
    cfg.update({"single_system_md_steps_per_s": 1200.0, "single_system_us_per_md_step": 800.0, "single_system_ms_per_pass": 16.0})
    out = bench._ordered_config(cfg)
    keys = list(out)
    assert keys[0] == "workload" and len(out["workload"]) <= 100 and out["workload_full"] == "w" * 140
    head = keys[:20]
    for name in ("lj4096", "water192", "schnet4096"):
        for suffix in ("_md_steps_per_s", "_ms_per_pass", "_kernel_frac"):
            assert name + suffix in head, (name + suffix, head)
    assert "single_system_md_steps_per_s" in head and "single_system_ms_per_pass" in head and "rdf_fused" in head
    assert keys.index("lj4096_kernel_frac") < keys.index("lj4096_parity_max_abs_dq") and keys.index("rdf_fused") < keys.index("loss")
    assert set(out) == set(cfg) | {"workload_full"}, "nothing is dropped"
    # the real headline's workload string fits the parser's cut without the fallback
    import inspect
    src = inspect.getsource(bench.run_lj108)
    assert "rep/GPU" in src
    assert len("FCC 3^3 %s 108 atoms rc 2.5 NHC(Q50,5) dt %g, %d steps+RDF loss+adjoint, %d rep/GPU"
               % ("ExVol(1,1,12)", 0.005, 49, 16384)) <= 100


def test_compulsory_bytes_of_the_fused_adjoint_launch():
    bench = importlib.import_module("bench")
    b = bench._lj108_compulsory_bytes(108, 16384, 50, 2)
    per_step_replica = b / (16384 * 49.0)
    assert 2500 < per_step_replica < 2800, "one (v, q, pv) frame of 108 atoms per step-replica: ~2.6 KB"
