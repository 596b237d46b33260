set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests -q -m gpu --tb=line 2>&1 | grep -E "passed|failed|FAILED|Error" | head -20
