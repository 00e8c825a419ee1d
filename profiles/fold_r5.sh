#!/bin/bash
# Fold gpurun_out/r5 (profiles/collect_r5.sh on the GPU box) into the tracked summaries profiles/r5_*.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r5
cp $O/stats/*/*_kernel_stats.csv profiles/r5_kernel_stats.csv
python profiles/make_timeline.py $O/stats/*/*_kernel_trace.csv profiles/r5_step_timeline.csv profiles/r5_step_timeline.md
python profiles/make_pmc_json.py profiles/r5_pmc_traffic_and_sq.json "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* each in its own run, --output-format csv, no tracing) of 'bench.py --no-cpu-baseline --no-f32-line --no-extra-configs --no-gather-bench --steps 3 --warmup 1' (round-5 defaults: fp16x3 arithmetic, k_out_fwd_h3p (wave pairs), k_out_dw_q with Adam + next-step operands, one-kernel head prefetched beside the dW kernel, bias operand in the Adam launch) on MI355X, config 2; per-dispatch means" $O/pmc_fetch $O/pmc_write $O/pmc_sq
for f in $O/bench_n1*.json; do tail -1 $f > profiles/r5_$(basename $f); done
for f in $O/ab_*.json; do tail -1 $f > "profiles/r5_$(basename $f | tr '=' '_')"; done
mkdir -p profiles/r5_ep profiles/r5_dp
for f in $O/bench_ep_*.json; do [ -s $f ] && tail -1 $f > profiles/r5_ep/$(basename $f); done
for f in $O/bench_dp_*.json; do [ -s $f ] && tail -1 $f > profiles/r5_dp/$(basename $f); done
[ -d $O/stats_dp8 ] && cp $O/stats_dp8/*/*_kernel_stats.csv profiles/r5_dp/kernel_stats_rank_of_8.csv
ls -la profiles/r5_*
