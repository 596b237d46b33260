set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_overlap.py tests/test_gpu_full_size.py tests/test_gpu_parity_at_size.py -q -m gpu --tb=short -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5
python tools/ab_bench.py DX_SMALL_DW_SIDE 0 1 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_dw.log
python tools/ab_bench.py DX_SMALL_DW_SIDE 0 1 -- --no-cpu-baseline 2>&1 | tee -a gpurun_out/r3/ab_dw.log
