set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r3/tests2.log
python bench.py --no-cpu-baseline > gpurun_out/r3/bench2.json 2> gpurun_out/r3/bench2.err
DX_ATTN_FUSED_BWD=0 python bench.py --no-cpu-baseline --no-probe > gpurun_out/r3/bench2_twopass.json 2>> gpurun_out/r3/bench2.err
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU"
rocprofv3 --kernel-trace --pmc $P1 -d gpurun_out/r3/pa1 -- python tools/bench_ops.py attn > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc $P2 -d gpurun_out/r3/pa2 -- python tools/bench_ops.py attn > /dev/null 2>&1
python tools/pmc_counters.py gpurun_out/r3/pa1 gpurun_out/r3/pa2 --out gpurun_out/r3/attn_counters.json --command "python tools/bench_ops.py attn" > /dev/null 2>&1
rm -rf gpurun_out/r3/pa1 gpurun_out/r3/pa2
cat gpurun_out/r3/tests2.log; head -c 600 gpurun_out/r3/bench2.json; echo; head -c 300 gpurun_out/r3/bench2_twopass.json
