set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python tools/ab_bench.py DX_WGRAD_HOLD_DEC 0 1 2 4 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_hold.log
python tools/ab_bench.py DX_WGRAD_HOLD_DEC 0 1 2 4 -- --no-cpu-baseline 2>&1 | tee -a gpurun_out/r3/ab_hold.log
