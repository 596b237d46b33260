set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py -q -m gpu --tb=short -x 2>&1 | tail -3
LP=$PWD/ubisoft-laforge-daft-exprt_amd/csrc/libdx_prev.so
for i in 1 2 3 4; do
DX_HIP_LIB=$LP python bench.py --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prev  ', d['ms_per_step'])"
python bench.py --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zmajor', d['ms_per_step'])"
done
