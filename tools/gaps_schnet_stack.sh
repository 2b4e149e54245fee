# Idle time between kernels in the steady state of the stacked SchNet pass (8 x 4 096 beads, rows16): rocprofv3 kernel trace of
# tools/gbench.py, gaps of the last <ms> of the trace.   gpurun -- bash tools/gaps_schnet_stack.sh [ms] [extra gbench args]
MS=${1:-100}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/q0
rocprofv3 --kernel-trace -d /tmp/q0 -o run -- python $R/tools/gbench.py gnn4096 --replicas 8 --steps 52 --bf16-rows "$@" > $O/gaps_stack_run.txt 2>&1
DB0=$(find /tmp/q0 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py gaps $DB0 40 $MS > $O/gaps_stack.txt 2>&1
python $R/tools/rocpd_summary.py stats $DB0 $MS > $O/stats_stack_window.txt 2>&1
grep -v simple_timer $O/gaps_stack_run.txt | tail -3; head -48 $O/stats_stack_window.txt | cut -c1-150; head -12 $O/gaps_stack.txt | cut -c1-160; sed -n '/idle time by/,$p' $O/gaps_stack.txt | head -45 | cut -c1-150
