export GPU_MAX_HW_QUEUES=16; mkdir -p gpurun_out/suite8
python -m pytest tests -q -m gpu -x > gpurun_out/suite8/gpu_suite.log 2>&1; tail -3 gpurun_out/suite8/gpu_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/suite8/bench.json 2> gpurun_out/suite8/bench.err
python scripts/print_bench_line.py gpurun_out/suite8/bench.json
