set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_conv.py -q -m gpu --tb=short -x 2>&1 | tail -3
L0=$PWD/ubisoft-laforge-daft-exprt_amd/csrc/libdx_sk0.so
for i in 1 2 3; do
DX_HIP_LIB=$L0 python bench.py --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spread0', d['ms_per_step'], d['roofline']['avg_launch_us'])"
python bench.py --no-cpu-baseline --steps 20 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('spread1', d['ms_per_step'], d['roofline']['avg_launch_us'])"
done
