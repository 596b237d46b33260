set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_conv.py -q -m gpu -k wide --tb=short 2>&1 | grep -E "^E |passed|failed" | head -20
for i in 1 2; do for w in 1 0; do echo -n "WIDE=$w: "; DX_CONV_WIDE=$w timeout 300 python bench.py --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"; done; done
echo -n "wgrad stream low priority: "; DX_WGRAD_PRIO=low timeout 300 python bench.py --no-cpu-baseline --no-probe 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3))"
python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -E "^E  |passed|failed|FAILED|^____" | head -60
