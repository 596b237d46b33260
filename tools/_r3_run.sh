# scratch script for gpurun calls during development (rewritten per run)
set -u
export TMPDIR=/tmp
bash tools/collect_profiles.sh r03 2>&1 | tail -3 | cut -c1-260
