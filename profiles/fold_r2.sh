#!/bin/bash
# Fold gpurun_out/r2 (profiles/collect_r2.sh on the GPU box) into the tracked summaries profiles/r2_*.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r2
cp $O/stats/*/*_kernel_stats.csv profiles/r2_kernel_stats.csv
python profiles/make_pmc_json.py profiles/r2_pmc_traffic_and_sq.json "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* each in its own run, --output-format csv, no tracing) of 'bench.py --no-cpu-baseline --no-f32-line --no-gather-bench --steps 3 --warmup 1' (defaults: fp16x3 arithmetic, k_out_fwd_h3w, k_out_dw_p2, flat Adam) on MI355X, config 2; per-dispatch means" $O/pmc_fetch $O/pmc_write $O/pmc_sq
python profiles/make_pmc_json.py profiles/r2_pmc_gather.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of 'bench.py --gather-only' (five whole-dataset launches of k_gather_pool: 1 995 708 teams, 46 MB table, d = 128); per-dispatch means" $O/pmc_gather_fetch $O/pmc_gather_write
for f in $O/bench_n1*.json; do tail -1 $f > profiles/r2_$(basename $f); done
mkdir -p profiles/r2_ep
tail -1 $O/bench_n1.json > profiles/r2_ep/bench_n1_same_box.json
for f in $O/bench_ep_*.json; do [ -s $f ] && tail -1 $f > profiles/r2_ep/$(basename $f); done
ls -la profiles/r2_*
