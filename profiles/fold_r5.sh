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

[ -d $O/d2v_pmc_fetch ] && python profiles/make_pmc_json.py profiles/r5_d2v_pmc.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_* (separate passes) of profiles/d2v_pass.py: three PV-DM passes (k_d2v_epoch<2, 5>: word-vector additions deferred to one per kept position) and one PV-DBOW pass (k_d2v_epoch<2, 0>) over the bench's dblp-shaped corpus; per-dispatch means per kernel" $O/d2v_pmc_fetch $O/d2v_pmc_write $O/d2v_pmc_tcc
[ -d $O/d2v_stats ] && cp $O/d2v_stats/*/*_kernel_stats.csv profiles/r5_d2v_kernel_stats.csv
[ -f $O/d2v_passes.txt ] && { echo "# default (PV-DM: deferred word-vector additions)"; grep "dm=" $O/d2v_passes.txt; echo "# NTF_D2V_DEFER=0 (round 4's kernel)"; grep "dm=" $O/d2v_passes_NTF_D2V_DEFER_0.txt; } > profiles/r5_d2v_passes.txt
ls -la profiles/r5_*
