set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests -q -m gpu --tb=short 2>&1 | tail -4 > gpurun_out/r3/full.log
cat gpurun_out/r3/full.log
bash tools/collect_profiles.sh r03 2>&1 | tail -7
