# round 4, GPU call 2: f4 kernels + the whole GPU suite on the restructured cfconv kernels + counters of the SchNet pass
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
(timeout 600 python -m pytest tests/test_gpu_bonded.py -m gpu -q -x 2>&1 | tail -25) > $O/c2_bonded.log; tail -4 $O/c2_bonded.log
(timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_bonded.py 2>&1 | tail -25) > $O/c2_pytest.log; tail -4 $O/c2_pytest.log
bash tools/prof_round3.sh r04a schnet4096 > $O/c2_prof.log 2>&1; tail -16 $O/c2_prof.log | cut -c1-160
