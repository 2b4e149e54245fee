# issue-side counters of the four headline kernels (two separate --pmc passes, kernel trace only)
set -x
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace -d /tmp/q1 -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SALU SQ_WAVES --kernel-trace -d /tmp/q2 -o run -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
A=$(find /tmp/q1 -name "*results.db" | head -1); B=$(find /tmp/q2 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py pmc $A $B | grep -i "traj_\|rdf_\|^#\|^kernel" > $R/gpurun_out/prof/pmc_issue.txt
wc -l $R/gpurun_out/prof/pmc_issue.txt
