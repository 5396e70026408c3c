#!/bin/bash
# L2 behaviour of the workgroup-per-QP kernel on BASELINE config 5: bytes fetched from beyond L2 and the L2 hit rate.
# One counter group per pass (FETCH_SIZE takes 3 of the 4 TCC slots); every pass under its own timeout.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
run() { name=$1; shift; timeout -k 5 240 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_c5_$name -- python3 bench.py --only config5 > gpurun_out/pmc_c5_$name.log 2>&1; echo "pass $name rc=$?"; }
run fetch FETCH_SIZE && run hit TCC_HIT_sum TCC_MISS_sum && run write WRITE_SIZE && \
python3 scripts/pmc_summary.py solve_block gpurun_out/pmc_c5_fetch gpurun_out/pmc_c5_hit gpurun_out/pmc_c5_write
