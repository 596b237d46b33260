set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_model.py tests/test_gpu_conv.py tests/test_gpu_full_size.py tests/test_gpu_generate.py -q -m gpu --tb=short -x 2>&1 | tail -5
python tools/ab_bench.py DX_BATCH_PREP 0 1 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_prep.log
