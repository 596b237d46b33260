set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python tools/ab_bench.py DX_MEL_BF16 0 1 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_mel.log
python tools/ab_bench.py DX_MEL_BF16 0 1 -- --no-cpu-baseline 2>&1 | tee -a gpurun_out/r3/ab_mel.log
