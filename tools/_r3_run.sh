set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_kernels.py -q -m gpu --tb=short -x -k attention 2>&1 | tail -3
LP=$PWD/ubisoft-laforge-daft-exprt_amd/csrc/libdx_prev.so
DX_HIP_LIB=$LP python tools/bench_ops.py attn 2>&1 | grep attn
python tools/bench_ops.py attn 2>&1 | grep attn
for i in 1 2 3; do
DX_HIP_LIB=$LP python bench.py --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev', d['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('xcd ', d['ms_per_step'])"
done
