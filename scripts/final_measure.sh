#!/bin/bash
# The measurement run behind profiles/r02_m..s: benches, kernel statistics and the separate --pmc passes, one MI355X.
set -e
# rocprofv3 initialises HIP before python starts: the queue request bench.py makes for itself has to be exported here
export GPU_MAX_HW_QUEUES=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r02_final_bench_default.json 2> gpurun_out/r02_final_bench_default.err
python bench.py --functional b3lyp --no-secondary --no-cpu-baseline > gpurun_out/r02_final_bench_b3lyp.json 2> gpurun_out/r02_final_b3lyp.err
python bench.py --df --no-secondary --no-cpu-baseline > gpurun_out/r02_final_bench_df.json 2> gpurun_out/r02_final_df.err
rocprofv3 --kernel-trace --stats -d gpurun_out/fprof_rhf -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/fprof_rhf.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/fprof_b3lyp -- python3 bench.py --functional b3lyp --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/fprof_b3lyp.log 2>&1
python3 scripts/rocpd_stats.py gpurun_out/fprof_rhf gpurun_out/r02_final_kernel_stats_rhf.csv 12
python3 scripts/rocpd_stats.py gpurun_out/fprof_b3lyp gpurun_out/r02_final_kernel_stats_b3lyp.csv 6
SQ="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/fpmc_fetch -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/fpmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/fpmc_write -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/fpmc_write.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/fpmc_sq -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/fpmc_sq.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/fpmc_sq_b3lyp -- python3 bench.py --functional b3lyp --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/fpmc_sq_b3lyp.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d gpurun_out/fpmc_sq_df -- python3 bench.py --df --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > gpurun_out/fpmc_sq_df.log 2>&1
python3 scripts/pmc_summary.py fetch=gpurun_out/fpmc_fetch write=gpurun_out/fpmc_write sq=gpurun_out/fpmc_sq sq_b3lyp=gpurun_out/fpmc_sq_b3lyp sq_df=gpurun_out/fpmc_sq_df > gpurun_out/r02_final_pmc_summary.txt
cp profiles/r02_pmc_*.csv profiles/r02_pmc_summary.json gpurun_out/
echo done
