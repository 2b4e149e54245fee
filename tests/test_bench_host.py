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
