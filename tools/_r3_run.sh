set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/profiles_r03
rm -rf /tmp/kt && rocprofv3 --kernel-trace -d /tmp/kt -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe > /tmp/kt.out 2> /tmp/kt.err
DB=$(find /tmp/kt -name "*.db" | head -1)
python $R/tools/rocprof_stats.py $DB > $R/gpurun_out/profiles_r03/r03_kernel_trace.md
python $R/tools/timeline.py $DB $R/gpurun_out/profiles_r03/r03_step_timeline.txt
