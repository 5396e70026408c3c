#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 -L 2>/dev/null | grep -i -E "ICACHE|IFETCH|SQ_INST_LEVEL|SQ_WAIT_INST_ANY|SQ_INSTS_VALU\b|LDS_UNALIGNED|SQ_LDS_IDX" | head -30 > gpurun_out/counters_avail.txt
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_$name -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$name.log 2>&1; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES
cat gpurun_out/counters_avail.txt
python3 scripts/pmc_summary.py gpurun_out/pmc_ic
tail -3 gpurun_out/pmc_ic.log
