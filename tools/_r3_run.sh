set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_full_size.py tests/test_gpu_overlap.py -q -m gpu 2>&1 | grep -E "^E |assert|Error|passed|failed" | head -60 > gpurun_out/r3/tests_a.log
python -m pytest "tests/test_gpu_parity_at_size.py::test_c5_long_utterance_batch_matches_oracle_slice" -q -m gpu -k bf16 2>&1 | grep -E "^E |assert|Error|passed|failed|stage-by-stage|^    [0-9]" | head -60 > gpurun_out/r3/tests_b.log
python -m pytest "tests/test_gpu_parity_at_size.py::test_c2_bench_batch_matches_oracle_slice" -q -m gpu -k bf16 -s 2>&1 | grep -E "stage-by-stage|^    [0-9]|passed|failed" | head -40 > gpurun_out/r3/tests_c.log
python -m pytest tests/test_gpu_training_curve.py -q -m gpu -s 2>&1 | grep -E "windowed|passed|failed|^E " | head > gpurun_out/r3/tests_d.log
python bench.py --loop train --steps 60 --warmup 20 2> gpurun_out/r3/bench_train_loop.err | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train() loop', round(d['ms_per_step'],3), round(d['value']), d['config']['valid_frames_per_step'])"
cat gpurun_out/r3/tests_a.log gpurun_out/r3/tests_b.log gpurun_out/r3/tests_c.log gpurun_out/r3/tests_d.log
