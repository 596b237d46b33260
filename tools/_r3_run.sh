# scratch script for gpurun calls during development (rewritten per run)
set -u
export TMPDIR=/tmp
python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -E "^E |passed|failed|rror|ERROR" | head
python bench.py --no-cpu-baseline --no-probe 2>/dev/null | cut -c1-200
