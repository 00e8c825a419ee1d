#!/bin/bash
# Fold gpurun_out/r6 (profiles/collect_r6.sh on the GPU box) into the tracked summaries profiles/r6_*.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r6
newest() { ls -t $1 | head -1; }      # (a re-collection merges beside the previous run's files: take the latest)
cp "$(newest "$O/stats/*/*_kernel_stats.csv")" profiles/r6_kernel_stats.csv
cp "$(newest "$O/stats_fnn/*/*_kernel_stats.csv")" profiles/r6_kernel_stats_fnn.csv
cp "$(newest "$O/stats_c3/*/*_kernel_stats.csv")" profiles/r6_kernel_stats_config3.csv
python profiles/make_timeline.py "$(newest "$O/stats/*/*_kernel_trace.csv")" profiles/r6_step_timeline.csv profiles/r6_step_timeline.md
python profiles/make_pmc_json.py profiles/r6_pmc_traffic_and_sq.json "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* each in its own run, --output-format csv, no tracing) of 'bench.py --no-cpu-baseline --no-f32-line --no-extra-configs --no-gather-bench --steps 3 --warmup 1' (round-6 defaults: fp16x3 arithmetic, k_out_fwd_h3p (wave pairs), k_out_dw_q with Adam + next-step operands, one-kernel head prefetched beside the dW kernel; the run's evaluation pass launches k_out_fwd_h3e) on MI355X, config 2; per-dispatch means" $O/pmc_fetch $O/pmc_write $O/pmc_sq
for f in $O/bench_n1*.json; do tail -1 $f > profiles/r6_$(basename $f); done
for f in $O/ab_*.json; do tail -1 $f > "profiles/r6_$(basename $f)"; done
ls -la profiles/r6_*
