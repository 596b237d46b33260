set -u
export TMPDIR=/tmp
for i in 1 2; do
for v in 0 1; do
DX_LN_GEMM2=$v DX_LNBWD_GEMM2=$v python bench.py --batch 256 --tmin 500 --steps 10 --warmup 15 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 gemm2=$v', round(d['ms_per_step'],3))"
DX_LN_GEMM2=$v DX_LNBWD_GEMM2=$v python bench.py --workload synth --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('synth gemm2=$v', round(d['ms_per_step'],3))"
done; done
python bench.py --no-cpu-baseline --no-probe 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2', round(d['ms_per_step'],3))"
