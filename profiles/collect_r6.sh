#!/bin/bash
# Round-6 profile collection on the GPU box:  /usr/local/graft/bin/gpurun --timeout 3300 -- 'bash profiles/collect_r6.sh'
# Outputs land in gpurun_out/r6/ ; the summaries are folded into profiles/r6_* by profiles/fold_r6.sh here afterwards.
# (Unchanged since round 4-5 and not re-collected: the -DNTF_DIAG stamps of the two output-layer kernels, the gather and doc2vec PMC passes, the power / clock samples,
#  the expert-shard and data-parallel one-rank emulations.)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-f32-line --no-extra-configs"
# bench lines (un-profiled)
python3 $R/bench.py --steps 50 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-f32-line --no-gather-bench > $O/bench_n1_driver_style.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 50 --warmup 10 --model fnn > $O/bench_n1_fnn.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --mfma f32 > $O/bench_n1_f32mfma.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 20 --warmup 3 --dataset dblp_full --rows 200000 > $O/bench_n1_dblp_full.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --input multihot --nsd unigram > $O/bench_n1_config3_multihot_unigram.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset uspt --d 256 > $O/bench_n1_config4_uspt_d256.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset gith > $O/bench_n1_config5_gith.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 50 --warmup 10 --nsd unigram_b > $O/bench_n1_nsd_unigram_b.json 2>> $O/bench.err
# A/B of this round's switches, same box, interleaved; the default line before and after them
python3 $B --no-gather-bench --steps 50 --warmup 10 > $O/ab_default_a.json 2>> $O/bench.err
NTF_EVAL_KERNEL=0 python3 $B --no-gather-bench --steps 50 --warmup 10 > $O/ab_NTF_EVAL_KERNEL_0.json 2>> $O/bench.err
NTF_FNN_PIPE=0 python3 $B --no-gather-bench --steps 50 --warmup 10 --model fnn > $O/ab_fnn_NTF_FNN_PIPE_0.json 2>> $O/bench.err
NTF_EVAL_KERNEL=0 python3 $B --no-gather-bench --steps 50 --warmup 10 --model fnn > $O/ab_fnn_NTF_EVAL_KERNEL_0.json 2>> $O/bench.err
NTF_L0_SWEEP=0 python3 $B --no-gather-bench --steps 30 --warmup 5 --input multihot --nsd unigram > $O/ab_config3_NTF_L0_SWEEP_0.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 50 --warmup 10 > $O/ab_default_b.json 2>> $O/bench.err
export NTF_BENCH_MIN_TIMED_S=0.01
# kernel trace (every dispatch: the step timeline) + stats of the default run, of the Fnn run and of config 3
rm -rf $O/stats; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --no-gather-bench --steps 200 --warmup 10 > $O/stats.log 2>&1
rm -rf $O/stats_fnn; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fnn -- python3 $B --no-gather-bench --steps 200 --warmup 10 --model fnn > $O/stats_fnn.log 2>&1
rm -rf $O/stats_c3; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_c3 -- python3 $B --no-gather-bench --steps 100 --warmup 10 --input multihot --nsd unigram > $O/stats_c3.log 2>&1
# PMC passes (each on its own, no tracing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null
# keep the merge-back under 64 MiB: the traces of the Fnn / config-3 runs are only wanted for their stats
rm -f $O/stats_fnn/*/*_kernel_trace.csv $O/stats_c3/*/*_kernel_trace.csv
du -sh $O | tail -1
