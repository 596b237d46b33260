#!/bin/bash
# build the library for gfx950 (cross-compiles here), then run a command on an MI355X box through gpurun (retrying while every GPU
# slot of the pod is busy -- nothing is charged for those attempts):
#   tools/gpu.sh <timeout_s> '<command>'
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" | tail -1 || exit 1
for attempt in $(seq 1 20); do
  out=$(/usr/local/graft/bin/gpurun --timeout "$1" -- "export TMPDIR=/tmp; $2" 2>&1)
  rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"
  exit $rc
done
echo "$out"; exit 3
