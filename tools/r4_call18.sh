cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; O=gpurun_out
python tools/hostprof_schnet.py --f32 --opt 12 > $O/c18_f32_opt.txt 2>&1; grep "^pass" $O/c18_f32_opt.txt
python tools/hostprof_schnet.py --opt 8 > $O/c18_bf16_opt.txt 2>&1; grep "^pass" $O/c18_bf16_opt.txt
(timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -30) > $O/c18_pytest.log; tail -3 $O/c18_pytest.log
