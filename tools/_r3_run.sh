set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_overlap.py -q -m gpu --tb=short -x 2>&1 | tail -3
python tools/ab_bench.py DX_WGRAD_DEFER_ROWS 0 16384 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_defer.log
python tools/ab_bench.py DX_WGRAD_DEFER_ROWS 0 16384 -- --no-cpu-baseline 2>&1 | tee -a gpurun_out/r3/ab_defer.log
