set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k "wgrad" 2>&1 | grep -E "^E |passed|failed|rror" | head
python -m pytest tests/test_gpu_model.py tests/test_gpu_ddp.py tests/test_gpu_overlap.py tests/test_gpu_full_size.py -q -m gpu --tb=short -x 2>&1 | grep -E "^E |passed|failed|rror" | head
python tools/ab_bench.py DX_WGRAD_PAIR 0 1 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_pair.log
python tools/ab_bench.py DX_WGRAD_PAIR 0 1 -- --no-cpu-baseline 2>&1 | tee -a gpurun_out/r3/ab_pair.log
