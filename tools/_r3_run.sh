set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_parity_at_size.py tests/test_gpu_full_size.py -q -m gpu --tb=short -x 2>&1 | tail -3
python tools/ab_bench.py DX_CONV_WFRAG 0 1 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_wfrag.log
python tools/ab_bench.py DX_CONV_WFRAG 0 1 -- --no-cpu-baseline 2>&1 | tee -a gpurun_out/r3/ab_wfrag.log
