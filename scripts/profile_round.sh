#!/bin/bash
# Round-end profiles: kernel-trace stats + PMC passes (separate runs; never combined with trace domains).
# Usage (on the GPU box): bash scripts/profile_round.sh <tag>
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_$tag* gpurun_out/pmc_${tag}_*      # (the merge back keeps older files of the same name otherwise)
# (long enough for the card to be past its clock ramp for most of the launches: DESIGN.md "How bench.py times")
BENCH="python3 bench.py --steps 200 --warmup 60 --no-cpu-baseline --no-closed-loop --no-extras"
# ---- the bench kernel (BASELINE configs[1])
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- $BENCH > gpurun_out/prof_$tag.json 2> gpurun_out/prof_$tag.err
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/pmc_${tag}_$name -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-closed-loop --no-extras > gpurun_out/pmc_${tag}_$name.log 2>&1; }
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT
run mfma SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
run fetch FETCH_SIZE
run write WRITE_SIZE
python3 scripts/pmc_summary.py solve_kernel gpurun_out/pmc_${tag}_sq1 gpurun_out/pmc_${tag}_sq2 gpurun_out/pmc_${tag}_mfma gpurun_out/pmc_${tag}_fetch gpurun_out/pmc_${tag}_write > gpurun_out/pmc_${tag}_summary.txt
cat gpurun_out/pmc_${tag}_summary.txt
find gpurun_out/prof_$tag -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-200 > gpurun_out/prof_${tag}_kernel_stats.csv
head -4 gpurun_out/prof_${tag}_kernel_stats.csv
# ---- configs[2] (N = 20 extended controller, batch 65536) and configs[4] (n = 12, m = 4, N = 30, batch 16384: block kernel, MFMA)
for cfg in config3 config5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_$cfg -- python3 bench.py --only $cfg > gpurun_out/prof_${tag}_$cfg.json 2> gpurun_out/prof_${tag}_$cfg.err
  find gpurun_out/prof_${tag}_$cfg -name "*kernel_stats.csv" -exec cat {} \; | cut -c1-200 > gpurun_out/prof_${tag}_${cfg}_kernel_stats.csv
  head -4 gpurun_out/prof_${tag}_${cfg}_kernel_stats.csv
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d gpurun_out/pmc_${tag}_${cfg}_a -- python3 bench.py --only $cfg > gpurun_out/pmc_${tag}_${cfg}_a.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_${tag}_${cfg}_b -- python3 bench.py --only $cfg > gpurun_out/pmc_${tag}_${cfg}_b.log 2>&1
  python3 scripts/pmc_summary.py solve_ gpurun_out/pmc_${tag}_${cfg}_a gpurun_out/pmc_${tag}_${cfg}_b > gpurun_out/pmc_${tag}_${cfg}_summary.txt
  cat gpurun_out/pmc_${tag}_${cfg}_summary.txt
done
# ---- the closed loop (tmpc_mc_run): the fused launch (closed_loop_kernel: all T steps of 4096 trajectories) and the launch pair per step, as bench.py times them
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${tag}_closed_loop -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/prof_${tag}_closed_loop.json 2> gpurun_out/prof_${tag}_closed_loop.err
python3 scripts/trace_summary.py gpurun_out/prof_${tag}_closed_loop > gpurun_out/summ_${tag}_closed_loop_kernel_trace_summary.txt
CL="python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_${tag}_cl_a -- $CL > gpurun_out/pmc_${tag}_cl_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/pmc_${tag}_cl_b -- $CL > gpurun_out/pmc_${tag}_cl_b.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${tag}_cl_f -- $CL > gpurun_out/pmc_${tag}_cl_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${tag}_cl_w -- $CL > gpurun_out/pmc_${tag}_cl_w.log 2>&1
python3 scripts/pmc_summary.py closed_loop_kernel gpurun_out/pmc_${tag}_cl_a gpurun_out/pmc_${tag}_cl_b gpurun_out/pmc_${tag}_cl_f gpurun_out/pmc_${tag}_cl_w > gpurun_out/summ_${tag}_closed_loop_pmc_summary.txt
# summaries written here, on the box, from this run's files only (what gets copied into profiles/)
python3 scripts/trace_summary.py gpurun_out/prof_$tag > gpurun_out/summ_${tag}_bench_kernel_trace_summary.txt
for cfg in config3 config5; do python3 scripts/trace_summary.py gpurun_out/prof_${tag}_$cfg > gpurun_out/summ_${tag}_${cfg}_kernel_trace_summary.txt; done
cp gpurun_out/pmc_${tag}_summary.txt gpurun_out/summ_${tag}_bench_pmc_summary.txt
for cfg in config3 config5; do cp gpurun_out/pmc_${tag}_${cfg}_summary.txt gpurun_out/summ_${tag}_${cfg}_pmc_summary.txt; done

