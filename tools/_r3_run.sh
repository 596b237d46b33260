# scratch script for gpurun calls during development (rewritten per run)
set -u
export TMPDIR=/tmp
python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -E "^E |passed|failed|rror|ERROR" | head
