cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_x_*
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 5 240 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_x_$c -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-closed-loop --no-extras > gpurun_out/pmc_x_$c.log 2>&1; done
python3 scripts/pmc_summary.py solve_kernel gpurun_out/pmc_x_FETCH_SIZE gpurun_out/pmc_x_WRITE_SIZE
