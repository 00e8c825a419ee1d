#!/bin/bash
# Fold gpurun_out/r4 (profiles/collect_r4.sh on the GPU box) into the tracked summaries profiles/r4_*.
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/r4
cp $O/stats/*/*_kernel_stats.csv profiles/r4_kernel_stats.csv
python profiles/make_timeline.py $O/stats/*/*_kernel_trace.csv profiles/r4_step_timeline.csv profiles/r4_step_timeline.md
python profiles/make_pmc_json.py profiles/r4_pmc_traffic_and_sq.json "rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_* each in its own run, --output-format csv, no tracing) of 'bench.py --no-cpu-baseline --no-f32-line --no-extra-configs --no-gather-bench --steps 3 --warmup 1' (round-4 defaults: fp16x3 arithmetic, k_out_fwd_h3p (wave pairs), k_out_dw_q with Adam + next-step operands and no f32 copy of sigma*eps, one-kernel head prefetched beside the dW kernel) on MI355X, config 2; per-dispatch means" $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_lds
[ -d $O/pmc_gather_fetch ] && python profiles/make_pmc_json.py profiles/r4_pmc_gather.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of 'bench.py --gather-only': five whole-dataset launches of k_gather_pool (1 995 708 teams, d = 128); per-dispatch means" $O/pmc_gather_fetch $O/pmc_gather_write
[ -d $O/d2v_pmc_fetch ] && python profiles/make_pmc_json.py profiles/r4_d2v_pmc.json "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_* (separate passes) of profiles/d2v_pass.py: three PV-DM passes and one PV-DBOW pass of k_d2v_epoch over the bench's dblp-shaped corpus; per-dispatch means over the four" $O/d2v_pmc_fetch $O/d2v_pmc_write $O/d2v_pmc_tcc
[ -d $O/d2v_stats ] && cp $O/d2v_stats/*/*_kernel_stats.csv profiles/r4_d2v_kernel_stats.csv && grep "dm=" $O/d2v_passes.txt > profiles/r4_d2v_passes.txt
for f in $O/bench_n1*.json; do tail -1 $f > profiles/r4_$(basename $f); done
for f in $O/ab_*.json; do tail -1 $f > "profiles/r4_$(basename $f | tr '=' '_')"; done
mkdir -p profiles/r4_ep
for f in $O/bench_ep_*.json; do [ -s $f ] && tail -1 $f > profiles/r4_ep/$(basename $f); done
cp $O/dw_stamps.txt profiles/r4_dw_stamps.txt
cp $O/fwd_stamps.txt profiles/r4_fwd_phase_stamps.txt
cp $O/fwd_pair_stamps.txt profiles/r4_fwd_pair_stamps.txt
for a in 0 2 3; do [ -s $O/fwd_h3x_abl_$a.json ] && tail -1 $O/fwd_h3x_abl_$a.json > profiles/r4_fwd_h3x_ablation_$a.json; done
[ -s $O/power_clocks.txt ] && cp $O/power_clocks.txt profiles/r4_power_clocks_samples.txt
[ -s $O/bench_long.json ] && tail -1 $O/bench_long.json > profiles/r4_bench_n1_20000_steps.json
python - <<'P'
import json, glob, os
O = "gpurun_out/r4"
def line(f):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r = {x["kernel"]: x for x in (d.get("roofline"), d.get("roofline_other")) if x}
    return d["ms_per_step"], r.get("out_fused_fwd_loss_dh", {}).get("avg_ms"), r.get("out_fused_dw_adam", {}).get("avg_ms")
rows = ["# Co-scheduling experiment (VERDICT r3 next #1d; `NTF_COSCHED` in a -DNTF_DIAG build; timing only - the results of these runs are garbage)", "",
        "The forward kernel on 8 x n persistent workgroups (one per CU, n column groups) while, on a second stream, the dW + Adam kernel (of the previous step's operands) takes the CUs it leaves free; the step's own dW launch is skipped.  `alone` = the same forward grid with no dW kernel in the step at all.", "",
        "| forward workgroups | forward alone ms | step, forward alone (no dW) ms | forward beside dW ms | dW beside forward ms | co-scheduled step ms |", "|---|---|---|---|---|---|"]
s = line(f"{O}/cosched_serial.json")
for n in (32, 28, 24, 20, 16, 12):
    a, b = line(f"{O}/cosched_fwd_alone_{n}.json"), line(f"{O}/cosched_beside_dw_{n}.json")
    rows.append(f"| {8 * n} | {a[1]:.3f} | {a[0]:.3f} | {b[1]:.3f} | {b[2]:.3f} | {b[0]:.3f} |")
rows += ["", f"Serial step on the same box, same build: {s[0]:.3f} ms (forward {s[1]:.3f}, dW + Adam {s[2]:.3f}).  Kill criterion: keep only if the step drops by >= 5 %."]
open("profiles/r4_cosched.md", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
P
ls -la profiles/r4_*
