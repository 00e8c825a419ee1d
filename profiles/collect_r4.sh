#!/bin/bash
# Round-4 profile collection on the GPU box:  /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash profiles/collect_r4.sh'
# Outputs land in gpurun_out/r4/ ; the summaries are folded into profiles/r4_* by profiles/fold_r4.sh here afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-f32-line --no-extra-configs"
# bench lines (un-profiled)
python3 $R/bench.py --steps 50 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-f32-line --no-gather-bench > $O/bench_n1_driver_style.json 2>> $O/bench.err
NTF_BENCH_NO_EVENTS=1 python3 $B --no-gather-bench --steps 50 --warmup 10 > $O/bench_n1_no_events.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 50 --warmup 10 --model fnn > $O/bench_n1_fnn.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --mfma f32 > $O/bench_n1_f32mfma.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 20 --warmup 3 --dataset dblp_full --rows 200000 > $O/bench_n1_dblp_full.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 20 --warmup 3 --dataset uspt_full --rows 200000 --d 256 > $O/bench_n1_uspt_full_d256.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --input multihot --nsd unigram > $O/bench_n1_config3_multihot_unigram.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset uspt --d 256 > $O/bench_n1_config4_uspt_d256.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset gith > $O/bench_n1_config5_gith.json 2>> $O/bench.err
# A/B of this round's changes, same box (each switch restores the round-3 behaviour of one piece); the default line before and after them
python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/ab_default_a.json 2>> $O/bench.err
for v in "NTF_FWD_KERNEL=3" "NTF_DW_KERNEL=0" "NTF_LEAN=0" "NTF_HEAD_PREFETCH=0" "NTF_PREFETCH=0" "NTF_HEAD=0" "NTF_SIDE_BWD=0" "NTF_FWD_KERNEL=4" "NTF_DW_STAGGER=0"; do
  env $v python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/ab_$v.json 2>> $O/bench.err
done
env NTF_FWD_KERNEL=3 NTF_DW_KERNEL=0 NTF_LEAN=0 NTF_HEAD_PREFETCH=0 python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/ab_round3_step_with_round4_adam.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/ab_default_b.json 2>> $O/bench.err
# expert-sharded multi-GPU path: what ONE rank of G runs, emulated on this GPU
for G in 2 4 8; do python3 $R/bench.py --steps 20 --warmup 4 --ep-emulate $G --no-extra-configs > $O/bench_ep_rank_of_$G.json 2>> $O/bench.err; done
# the -DNTF_DIAG part (stamps, co-scheduling experiment): profiles/collect_r4_diag.sh
bash $R/profiles/collect_r4_diag.sh
# package power and shader clock while the step runs (rocm-smi every 2 s beside a 20 000-step run)
python3 $B --no-gather-bench --steps 20000 --warmup 10 > $O/bench_long.json 2>> $O/bench.err &
BP=$!
rocm-smi --showmaxpower 2>/dev/null | grep "GPU\[" > $O/power_clocks.txt
for n in $(seq 1 40); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' ' >> $O/power_clocks.txt; echo >> $O/power_clocks.txt
  sleep 2
done
wait $BP
export NTF_BENCH_MIN_TIMED_S=0.01
# kernel trace (every dispatch: the step timeline) + stats of the default run
# (200 steps: the first ~15 dispatches of a fresh process run 5-40 % long - clocks and caches settling - and would carry a 23-dispatch average 5 % over the bench's)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --no-gather-bench --steps 200 --warmup 10 > $O/stats.log 2>&1
# PMC passes (each on its own, no tracing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_lds -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_lds.log 2>&1
if [ -z "$R4_SKIP_UNCHANGED" ]; then
# the whole-dataset gather launch (roofline_gather's traffic)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_gather_fetch -- python3 $B --gather-only --steps 1 --warmup 0 > $O/pmc_gather_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_gather_write -- python3 $B --gather-only --steps 1 --warmup 0 > $O/pmc_gather_write.log 2>&1
# the doc2vec passes: kernel stats and traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d2v_stats -- python3 $R/profiles/d2v_pass.py > $O/d2v_passes.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/d2v_pmc_fetch -- python3 $R/profiles/d2v_pass.py > $O/d2v_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/d2v_pmc_write -- python3 $R/profiles/d2v_pass.py > $O/d2v_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d $O/d2v_pmc_tcc -- python3 $R/profiles/d2v_pass.py > $O/d2v_pmc_tcc.log 2>&1
fi
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $O | tail -1
