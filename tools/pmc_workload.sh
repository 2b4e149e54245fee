# issue-side counters of one bench workload: pmc_workload.sh <workload> <tag> <kernel-grep> [extra bench args]
W=$1; TAG=$2; PAT=$3; shift 3
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/w1 /tmp/w2
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace -d /tmp/w1 -o run -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY --kernel-trace -d /tmp/w2 -o run -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
A=$(find /tmp/w1 -name "*results.db" | head -1); B=$(find /tmp/w2 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py pmc $A $B | grep -i "$PAT\|^#\|^kernel" > $R/gpurun_out/prof/${TAG}_${W}_pmc_issue.txt
cut -c1-40,70-140 $R/gpurun_out/prof/${TAG}_${W}_pmc_issue.txt | grep -v "^#" | head -60
