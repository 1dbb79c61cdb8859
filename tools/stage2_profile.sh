#!/bin/bash
# Kernel breakdown of ONE 2nd-stage PPO update (32 envs, M = 300) + the bench record: bash tools/stage2_profile.sh [out_dir]
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=${1:-$R/gpurun_out/stage2}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p2
python3 $R/bench.py --stage 2 --envs 32 --steps 2 --no-roofline --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > $O/bench_stage2_envs32.log || exit 1
rocprofv3 --kernel-trace --stats -d /tmp/p2 -o res -- python3 $R/bench.py --stage 2 --envs 32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras > /tmp/p2.log 2>&1 || exit 1
( echo "# 2nd stage (BASELINE configs[3] per-GPU share: 32 envs, M = 300 memory history), bf16x3: kernels of ONE PPO update (2 x 2)"; echo; echo '`rocprofv3 --kernel-trace --stats -- python bench.py --stage 2 --envs 32 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-extras`, everything after the last `gae_kernel`:'; echo; python3 $R/tools/update_breakdown.py $(find /tmp/p2 -name "*.db" | head -1) ) > $O/update_stage2.md || exit 1
python3 $R/tools/update_sequence.py $(find /tmp/p2 -name "*.db" | head -1) > $O/update_sequence.txt
cut -c1-400 $O/bench_stage2_envs32.log
head -40 $O/update_stage2.md
python3 $R/tools/step_trace.py $(find /tmp/p2 -name "*.db" | head -1) 150 40 > $O/step_trace_stage2.txt 2>&1
