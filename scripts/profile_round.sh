#!/bin/bash
# Round-end profile of the bench command: kernel-trace stats + PMC passes (separate runs).
# Usage (on the GPU box): bash scripts/profile_round.sh <tag>
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-closed-loop > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-closed-loop > gpurun_out/pmc_${tag}_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_summary.py gpurun_out/pmc_${tag}_sq1 gpurun_out/pmc_${tag}_sq2 gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write > gpurun_out/pmc_${tag}_summary.txt
cat gpurun_out/pmc_${tag}_summary.txt
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-260
