#!/bin/bash
# Collect the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   bash tools/collect_profiles.sh r02      -> gpurun_out/profiles_r02/*  (copy what is to be judged into profiles/)
# kernel trace + step timeline, three separate PMC passes (FETCH_SIZE / WRITE_SIZE / MFMA busy: they do not fit the TCC
# slots together, and gpurun refuses --pmc combined with the other trace domains), bench lines of C2 / C5 / C4.
set -u
TAG=${1:-r06}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
CMD="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-probe --no-secondary"
python bench.py --batch 256 --tmin 500 --steps 10 --warmup 15 --no-cpu-baseline --no-secondary > $OUT/${TAG}_bench_train_c5.json 2> $OUT/bench.err
rocprofv3 --kernel-trace -d $OUT/kt -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-probe --no-secondary > /dev/null 2> $OUT/kt.err
DB=$(ls $OUT/kt/*/*.db | head -1)
python tools/rocprof_stats.py $DB > $OUT/${TAG}_kernel_trace.md
python tools/timeline.py $DB $OUT/${TAG}_step_timeline.txt > /dev/null
rm -rf $OUT/kt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -- $CMD > /dev/null 2> $OUT/pmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -- $CMD > /dev/null 2> $OUT/pmc2.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $OUT/pmc_mfma -- $CMD > /dev/null 2> $OUT/pmc3.err
python tools/pmc_counters.py $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma --out $OUT/${TAG}_counters.json --command "$CMD" --steps-total 5 > $OUT/counters_summary.txt
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_mfma
# synthesis path (BASELINE configs[3]): kernel trace + the same three PMC passes
SCMD="python bench.py --workload synth --batch 256 --steps 3 --warmup 2 --no-cpu-baseline --no-probe"
rocprofv3 --kernel-trace -d $OUT/skt -- python bench.py --workload synth --batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-probe > /dev/null 2> $OUT/skt.err
SDB=$(ls $OUT/skt/*/*.db | head -1)
python tools/rocprof_stats.py $SDB > $OUT/${TAG}_synth_kernel_trace.md
rm -rf $OUT/skt
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/spmc_fetch -- $SCMD > /dev/null 2> $OUT/spmc1.err
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/spmc_write -- $SCMD > /dev/null 2> $OUT/spmc2.err
timeout 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d $OUT/spmc_mfma -- $SCMD > /dev/null 2> $OUT/spmc3.err
python tools/pmc_counters.py $OUT/spmc_fetch $OUT/spmc_write $OUT/spmc_mfma --out $OUT/${TAG}_synth_counters.json --command "$SCMD" --steps-total 5 > $OUT/synth_counters_summary.txt
rm -rf $OUT/spmc_fetch $OUT/spmc_write $OUT/spmc_mfma
P1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_INSTS_SALU"
timeout 300 rocprofv3 --kernel-trace --pmc $P1 -d $OUT/pa1 -- python tools/bench_ops.py attn > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $P2 -d $OUT/pa2 -- python tools/bench_ops.py attn > /dev/null 2>&1
python tools/pmc_counters.py $OUT/pa1 $OUT/pa2 --out $OUT/${TAG}_attention_counters.json --command "python tools/bench_ops.py attn" > /dev/null
rm -rf $OUT/pa1 $OUT/pa2
# the bench lines of C2 / C4 LAST: bench.py quotes roofline.traffic / hbm from profiles/<tag>_counters.json only when that file was
# recorded for the kernel build that is running, so the counters of THIS build go into profiles/ (of this box's copy) first
cp $OUT/${TAG}_counters.json $OUT/${TAG}_synth_counters.json profiles/
python bench.py > $OUT/${TAG}_bench.json 2>> $OUT/bench.err
python bench.py --workload synth --batch 256 --steps 10 --warmup 3 > $OUT/${TAG}_bench_synth_c4.json 2>> $OUT/bench.err
# the multi-rank code path on real RCCL through a one-rank world: what the process group, the per-bucket all-reduce
# hooks and the per-bucket optimizer cost before any byte crosses xGMI
DX_FORCE_DIST=1 python bench.py --no-secondary --no-cpu-baseline 2> $OUT/bench_rccl.err | grep "^{" > $OUT/${TAG}_bench_one_rank_rccl.json
grep "bench rank" $OUT/bench_rccl.err > $OUT/${TAG}_one_rank_rccl_reducer.txt
# the CPU oracle on the WHOLE B = 48 bench batch, 1 warm-up + 5 timed steps (the default run keeps its 8-utterance sample)
python bench.py --cpu-utts 48 --cpu-steps 5 --no-secondary --no-probe --steps 5 2>> $OUT/bench.err | python -c "import sys, json; d = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(json.dumps(d['cpu_baseline'], indent=1))" > $OUT/${TAG}_cpu_baseline_full_batch.json
cat $OUT/counters_summary.txt
head -c 400 $OUT/${TAG}_bench.json; echo
