# issue-side counters of the fused SchNet kernels (two separate --pmc passes, kernel trace only): TAG = prefix
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c1 /tmp/c2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS --kernel-trace -d /tmp/c1 -o run -- python $R/tools/kbench_cfconv.py --reps 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --kernel-trace -d /tmp/c2 -o run -- python $R/tools/kbench_cfconv.py --reps 3 > /dev/null 2>&1
A=$(find /tmp/c1 -name "*results.db" | head -1); B=$(find /tmp/c2 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py pmc $A $B | grep -i "cfconv\|dense_k\|^#\|^kernel" > $R/gpurun_out/prof/${TAG}_cfconv_pmc_issue.txt
cat $R/gpurun_out/prof/${TAG}_cfconv_pmc_issue.txt | cut -c1-140
