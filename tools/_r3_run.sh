# scratch script for gpurun calls during development (rewritten per run)
set -u
export TMPDIR=/tmp
python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "splitk or lnbwd" 2>&1 | grep -E "^E |passed|failed|rror|ERROR" | head
python -m pytest tests/test_gpu_model.py tests/test_gpu_full_size.py -q -m gpu --tb=short -x 2>&1 | grep -E "^E |passed|failed|rror|ERROR" | head
python tools/ab_bench.py DX_CONV_SPLITK_K1 0 1 -- --no-cpu-baseline 2>&1
python tools/ab_bench.py DX_CONV_SPLITK_K1 0 1 -- --no-cpu-baseline 2>&1
