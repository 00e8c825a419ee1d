#!/bin/bash
# Fold gpurun_out/r3 (profiles/collect_r3.sh on the GPU box) into the tracked summaries profiles/r3_*.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r3
cp $O/stats/*/*_kernel_stats.csv profiles/r3_kernel_stats.csv
python profiles/make_timeline.py $O/stats/*/*_kernel_trace.csv profiles/r3_step_timeline.csv profiles/r3_step_timeline.md
python profiles/make_pmc_json.py profiles/r3_pmc_traffic_and_sq.json "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* each in its own run, --output-format csv, no tracing) of 'bench.py --no-cpu-baseline --no-f32-line --no-extra-configs --no-gather-bench --steps 3 --warmup 1' (defaults: fp16x3 arithmetic, k_out_fwd_h3x, k_out_dw_p2 with Adam + next-step operands, one-kernel head) on MI355X, config 2; per-dispatch means" $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
for f in $O/bench_n1*.json; do tail -1 $f > profiles/r3_$(basename $f); done
for f in $O/ab_*.json; do tail -1 $f > "profiles/r3_$(basename $f | tr '=' '_')"; done
mkdir -p profiles/r3_ep
for f in $O/bench_ep_*.json; do [ -s $f ] && tail -1 $f > profiles/r3_ep/$(basename $f); done
cp $O/fwd_stamps.txt profiles/r3_fwd_phase_stamps.txt
ls -la profiles/r3_*
cat $O/fwd_stamps_repeat10.txt >> profiles/r3_fwd_phase_stamps.txt 2>/dev/null || true
[ -s $O/power_clocks.txt ] && cp $O/power_clocks.txt profiles/r3_power_clocks_samples.txt
[ -s $O/bench_long.json ] && tail -1 $O/bench_long.json > profiles/r3_bench_n1_20000_steps.json
