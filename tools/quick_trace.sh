#!/bin/bash
# kernel trace + step timeline of the headline step only (development): bash tools/quick_trace.sh <tag>
TAG=${1:-dev}
OUT=gpurun_out/trace_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT/kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-secondary > /dev/null 2> $OUT/kt.err
DB=$(ls $OUT/kt/*/*.db | head -1)
python tools/rocprof_stats.py $DB > $OUT/${TAG}_kernel_trace.md
python tools/timeline.py $DB $OUT/${TAG}_step_timeline.txt > /dev/null
rm -rf $OUT/kt
