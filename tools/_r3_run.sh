# scratch script for gpurun calls during development (rewritten per run):
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/_r3_run.sh'
set -u
export TMPDIR=/tmp
python -m pytest tests -q -m gpu --tb=short 2>&1 | grep -E "^E |passed|failed|rror|ERROR" | head
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -1
python bench.py 2>/dev/null | cut -c1-200
