set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
for i in 1 2 3; do
python -m pytest tests/test_gpu_parity_at_size.py -q -m gpu --tb=short -x -k "c5 and bf16" 2>&1 | grep -E "^E |passed|failed|assert" | head -12
done
