#!/bin/bash
# Collects the round's judged artifacts on the GPU box into gpurun_out/prof_final/ (copied into profiles/ afterwards):
# default bench line, GPU parity-test log, rocprofv3 kernel statistics of a 30-step rollout + update, per-step / update
# breakdowns, step timeline, and the three separate --pmc passes over tools/roofline_probe.py.
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_default.log 2>&1 || exit 1
tail -1 $O/bench_default.log | cut -c1-300
python3 -m pytest $R/tests -m gpu -q > $O/gpu_parity_tests.log 2>&1 || exit 1
tail -1 $O/gpu_parity_tests.log
rocprofv3 --kernel-trace --stats -d /tmp/ps -o res -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/ps.log 2>&1 || exit 1
DB=$(find /tmp/ps -name "*.db" | head -1)
python3 $R/tools/prof_summary.py $DB > $O/rocprof_summary_head.md || exit 1
python3 $R/tools/step_breakdown.py $DB 150 > $O/step_breakdown.md || exit 1
python3 $R/tools/update_breakdown.py $DB > $O/update_breakdown.md || exit 1
python3 $R/tools/step_timeline.py 64 fresh 2>&1 | grep -E "^N=|median|sequencer|dialog stats" > $O/step_timeline.txt || exit 1
python3 $R/tools/step_timeline.py 64 reference 2>&1 | grep -E "^N=|median|sequencer|dialog stats" >> $O/step_timeline.txt || exit 1
python3 $R/tools/launch_cost.py 2>&1 | grep -E "phase|GRAPH|RECORD|WAIT|MULTICOPY|launch:|us$" > $O/launch_cost.txt || exit 1
if [ -x $R/tools/bin/chain_lab1 ]; then ( echo "compensated-bf16 chain, pi_q-like program (16 Linear steps), warm / cold:"; $R/tools/bin/chain_lab1 1 | grep kernel ) > $O/chain_warm_cold.txt || exit 1; fi
python3 $R/tools/step_trace.py $DB 150 40 > $O/step_trace.txt || exit 1
# phase tables of the persistent tower launch and of the one-launch text tower (lab binaries built here: tools/bin is not tracked)
if [ -x $R/tools/bin/x3_lab ]; then $R/tools/bin/x3_lab 64 6 > $O/tower_x3_phases.txt || exit 1; fi
if [ -x $R/tools/bin/clip_lab ]; then
  # the lab build carries phase timers (cycle counts added to a table as each phase ends: same speed and, launch for launch, the
  # same bits as the product build -- the output checksum is printed)
  $R/tools/bin/clip_lab 64 0 0 1 0 > $O/clip_tower_phases.txt || exit 1
  $R/tools/bin/clip_lab 64 >> $O/clip_tower_phases.txt || exit 1
  $R/tools/bin/clip_lab 16 >> $O/clip_tower_phases.txt || exit 1
fi
python3 $R/tools/clip_det_probe.py 2>&1 | grep -E "n=|4-way" > $O/clip_tower_splits.txt || exit 1
python3 $R/tools/clip_xproc_probe.py 2>&1 | grep -E "^n=" >> $O/clip_tower_splits.txt || exit 1     # digests: equal from process to process
for n in 64 32 16 8; do python3 $R/tools/clip_time.py $n 2>&1 | grep "per call" >> $O/clip_tower_splits.txt; done
python3 $R/tools/clip_stream_probe.py 2>&1 | grep "stream=" > $O/clip_tower_parity_and_time.txt || exit 1
if [ -x $R/tools/bin/gru_seq_lab ]; then $R/tools/bin/gru_seq_lab 150 8 > $O/gru_seq_phases.txt || exit 1; fi
if [ -x $R/tools/bin/seq_lab ]; then timeout -k 10 120 $R/tools/bin/seq_lab 64 4096 > $O/handoff_latency_probe.txt || exit 1; fi
if [ -x $R/tools/bin/head_lab0 ]; then $R/tools/bin/head_lab0 64 6 128 1 > $O/tower_head_phases.txt || exit 1; fi
if [ -x $R/tools/bin/tail_lab ]; then $R/tools/bin/tail_lab 64 6 > $O/tower_tail_phases.txt || exit 1; fi
if [ -f $O/tower_head_phases.txt ] && [ -f $O/tower_tail_phases.txt ]; then python3 $R/tools/tower_phase_table.py $O/tower_head_phases.txt $O/tower_tail_phases.txt > $O/tower_phase_table.md || exit 1; fi
if [ -x $R/tools/bin/gridbar_lab ]; then $R/tools/bin/gridbar_lab 256 200 1 > $O/grid_barrier_probe.txt || exit 1; fi
AVLEN_CLIP_STREAM=0 python3 $R/tools/text_split_probe.py 2>&1 | grep -E "us$|tokens" > $O/text_tower_scaling.txt || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -o res -- python3 $R/tools/roofline_probe.py > /tmp/pmc_$c.log 2>&1 || exit 1
  cp $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $O/pmc_${c}_counter_collection.csv || exit 1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -o res -- python3 $R/tools/roofline_probe.py > /tmp/pmc_mfma.log 2>&1 || exit 1
cp $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) $O/pmc_MFMA_counter_collection.csv || exit 1
python3 $R/tools/pmc_traffic.py $O/pmc_FETCH_SIZE_counter_collection.csv $O/pmc_WRITE_SIZE_counter_collection.csv $O/pmc_traffic.json $O/pmc_MFMA_counter_collection.csv > /dev/null || exit 1
cat $O/pmc_traffic.json | head -30
# the other configurations (one line each)
python3 $R/bench.py --config gru --steps 3 --no-roofline --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_gru.log
python3 $R/bench.py --distractor --steps 3 --no-roofline --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_distractor.log
python3 $R/bench.py --stage 2 --envs 32 --steps 2 --no-roofline --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_stage2_envs32.log
python3 $R/bench.py --stage 2 --envs 32 --steps 2 --no-roofline --no-cpu-baseline --no-extras --precision bf16 2>/dev/null | tail -1 > $O/bench_stage2_envs32_bf16.log
rocprofv3 --kernel-trace --stats -d /tmp/p2 -o res -- python3 $R/bench.py --stage 2 --envs 32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/p2.log 2>&1 || exit 1
( echo "# 2nd stage (BASELINE configs[3] per-GPU share: 32 envs, M = 300 memory history), bf16x3: kernels of ONE PPO update (2 x 2)"; echo; echo '`rocprofv3 --kernel-trace --stats -- python bench.py --stage 2 --envs 32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras`, everything after the last `gae_kernel`:'; echo; python3 $R/tools/update_breakdown.py $(find /tmp/p2 -name "*.db" | head -1) ) > $O/update_stage2.md || exit 1
python3 $R/tools/update_sequence.py $(find /tmp/p2 -name "*.db" | head -1) > $O/update_sequence.txt || exit 1
python3 $R/tools/step_trace.py $(find /tmp/p2 -name "*.db" | head -1) 150 40 > $O/step_trace_stage2.txt 2>&1
python3 $R/tools/gemm_tn_time.py 2>&1 | grep -E "dW|max" > $O/gemm_tn_time.txt || exit 1
python3 $R/tools/audio3_ab.py 2>&1 | grep fused > $O/audio3_ab.txt || exit 1
python3 $R/tools/belief_ab.py 2>&1 | grep -E "env-steps" > $O/belief_ab.txt || exit 1
python3 $R/bench.py --precision bf16 --steps 3 --no-roofline --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_bf16.log
python3 $R/bench.py --belief --spectrogram 65x26 --steps 3 --no-roofline --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_belief_65x26.log
rocprofv3 --kernel-trace --stats -d /tmp/pg -o res -- python3 $R/bench.py --config gru --steps 1 --warmup 1 --no-cpu-baseline --no-roofline > /tmp/pg.log 2>&1 || exit 1
python3 $R/tools/prof_summary.py $(find /tmp/pg -name "*.db" | head -1) "rocprofv3 --kernel-trace --stats -- python bench.py --config gru --steps 1 --warmup 1" > $O/rocprof_gru.md
cut -c1-200 $O/bench_gru.log $O/bench_distractor.log $O/bench_stage2_envs32.log $O/bench_stage2_envs32_bf16.log $O/bench_bf16.log $O/bench_belief_65x26.log
