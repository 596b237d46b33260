#!/bin/bash
# build the library for gfx950 (cross-compiles here), then run a command on an MI355X box through gpurun:
#   tools/gpu.sh <timeout_s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" | tail -1
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "export TMPDIR=/tmp; $2"
