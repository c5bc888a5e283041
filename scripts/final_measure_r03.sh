#!/bin/bash
# The measurement run behind profiles/r03_*: benches, kernel statistics and the separate --pmc passes, one MI355X.
# Every line of the same call runs on the same box (box-to-box spread on this pool is ~ +-15 %).
set -e
export GPU_MAX_HW_QUEUES=16       # rocprofv3 initialises HIP before python starts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03_final}
mkdir -p $O
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --functional b3lyp --no-secondary --no-cpu-baseline > $O/bench_b3lyp.json 2> $O/bench_b3lyp.err
python bench.py --df --no-secondary --no-cpu-baseline > $O/bench_df.json 2> $O/bench_df.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rhf -o rhf -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_rhf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b3lyp -o b3lyp -- python3 bench.py --functional b3lyp --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_b3lyp.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_df -o df -- python3 bench.py --df --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/prof_df.log 2>&1
cp $O/prof_rhf/rhf_kernel_stats.csv $O/kernel_stats_rhf.csv; cp $O/prof_b3lyp/b3lyp_kernel_stats.csv $O/kernel_stats_b3lyp.csv; cp $O/prof_df/df_kernel_stats.csv $O/kernel_stats_df.csv
rm -rf $O/prof_rhf $O/prof_b3lyp $O/prof_df
SQ="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
B="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_b3lyp -- $B --functional b3lyp > $O/pmc_sq_b3lyp.log 2>&1
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $O/pmc_sq_df -- $B --df > $O/pmc_sq_df.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_df -- $B --df > $O/pmc_fetch_df.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_df -- $B --df > $O/pmc_write_df.log 2>&1
python3 scripts/pmc_summary.py fetch=$O/pmc_fetch write=$O/pmc_write sq=$O/pmc_sq sq_b3lyp=$O/pmc_sq_b3lyp sq_df=$O/pmc_sq_df fetch_df=$O/pmc_fetch_df write_df=$O/pmc_write_df > $O/pmc_summary.txt
cp profiles/r03_pmc_*.csv profiles/r03_pmc_summary.json $O/
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_sq_b3lyp $O/pmc_sq_df $O/pmc_fetch_df $O/pmc_write_df
echo done
