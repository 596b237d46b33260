# scratch script for gpurun calls during development (rewritten per run)
set -u
export TMPDIR=/tmp
python -m pytest tests -q -m gpu --tb=short -k "mel or frontend or feature or generate" 2>&1 | grep -E "^E |passed|failed|rror|ERROR" | head
python tools/bench_ops.py mel 2>&1 | grep "mel front-end"
