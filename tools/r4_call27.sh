cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6) > $O/f_pytest.log; tail -3 $O/f_pytest.log
(timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/f_smoke.log; tail -1 $O/f_smoke.log
