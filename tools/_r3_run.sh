set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python tools/ab_bench.py DX_WGRAD_BLOCKS 192 256 384 128 -- --no-cpu-baseline 2>&1 | tee gpurun_out/r3/ab_blocks.log
