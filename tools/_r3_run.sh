set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/profiles_r03
python bench.py > gpurun_out/profiles_r03/r03_bench.json 2> gpurun_out/profiles_r03/bench.err
python bench.py --workload synth --batch 256 --steps 10 --warmup 3 > gpurun_out/profiles_r03/r03_bench_synth_c4.json 2>> gpurun_out/profiles_r03/bench.err
python bench.py --loop train --steps 60 --warmup 20 2>/dev/null | tail -1 | cut -c1-300
