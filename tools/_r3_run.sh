set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_kernels.py -x -q -k attention 2>&1 | grep -E "passed|failed|Error|error" | tail -5
ATTN_ORDER=sorted,random python tools/bench_ops.py attn 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-probe 2>/dev/null | head -c 300; echo
