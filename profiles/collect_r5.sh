#!/bin/bash
# Round-5 profile collection on the GPU box:  /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash profiles/collect_r5.sh'
# Outputs land in gpurun_out/r5/ ; the summaries are folded into profiles/r5_* by profiles/fold_r5.sh here afterwards.
# (Unchanged since round 4 and not re-collected: the -DNTF_DIAG stamps of the two output-layer kernels, the gather and doc2vec PMC passes, the power / clock samples.)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-f32-line --no-extra-configs"
# bench lines (un-profiled)
python3 $R/bench.py --steps 50 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err
python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-configs --no-f32-line --no-gather-bench > $O/bench_n1_driver_style.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 50 --warmup 10 --model fnn > $O/bench_n1_fnn.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --mfma f32 > $O/bench_n1_f32mfma.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 20 --warmup 3 --dataset dblp_full --rows 200000 > $O/bench_n1_dblp_full.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 20 --warmup 3 --dataset uspt_full --rows 200000 --d 256 > $O/bench_n1_uspt_full_d256.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --input multihot --nsd unigram > $O/bench_n1_config3_multihot_unigram.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset uspt --d 256 > $O/bench_n1_config4_uspt_d256.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset gith > $O/bench_n1_config5_gith.json 2>> $O/bench.err
# the samplers the reference's committed runs use (all 40 committed Bnn directories are nsdunigram_b): per-batch table now staged beside the dW kernel, and the A/B without
python3 $B --no-gather-bench --steps 50 --warmup 10 --nsd unigram_b > $O/bench_n1_nsd_unigram_b.json 2>> $O/bench.err
NTF_HEAD_PREFETCH=0 python3 $B --no-gather-bench --steps 50 --warmup 10 --nsd unigram_b > $O/ab_unigram_b_NTF_HEAD_PREFETCH_0.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 50 --warmup 10 --nsd unigram > $O/bench_n1_nsd_unigram.json 2>> $O/bench.err
# A/B of this round's switches, same box; the default line before and after them
python3 $B --no-gather-bench --steps 50 --warmup 10 > $O/ab_default_a.json 2>> $O/bench.err
# (NTF_FIX_IN_FWD=1 was an arm of the first collections: profiles/r5_removed_experiment_NTF_FIX_IN_FWD_1.json; the experiment measured slower and its code is gone)
for v in "NTF_MERGE_BIAS=0" "NTF_HEAD_PREFETCH=0" "NTF_DW_KERNEL=0" "NTF_FWD_KERNEL=3"; do
  env $v python3 $B --no-gather-bench --steps 50 --warmup 10 > $O/ab_$v.json 2>> $O/bench.err
done
python3 $B --no-gather-bench --steps 50 --warmup 10 > $O/ab_default_b.json 2>> $O/bench.err
# multi-GPU forms: what ONE rank of G runs, emulated on this GPU (no exchange): expert-sharded and data-parallel (north_star's form)
for G in 2 4 8; do python3 $R/bench.py --steps 20 --warmup 4 --ep-emulate $G --no-extra-configs > $O/bench_ep_rank_of_$G.json 2>> $O/bench.err; done
for G in 2 4 8; do python3 $R/bench.py --steps 30 --warmup 5 --dp-emulate $G > $O/bench_dp_rank_of_$G.json 2>> $O/bench.err; done
NTF_DP_RANGES=0 python3 $R/bench.py --steps 30 --warmup 5 --dp-emulate 8 > $O/bench_dp_rank_of_8_NTF_DP_RANGES_0.json 2>> $O/bench.err
NTF_DP_SIDE_BWD=0 python3 $R/bench.py --steps 30 --warmup 5 --dp-emulate 8 > $O/bench_dp_rank_of_8_NTF_DP_SIDE_BWD_0.json 2>> $O/bench.err
export NTF_BENCH_MIN_TIMED_S=0.01
# kernel trace (every dispatch: the step timeline) + stats of the default run
rm -rf $O/stats; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --no-gather-bench --steps 200 --warmup 10 > $O/stats.log 2>&1
# ... and of one data-parallel rank of 8 (the ranged forward launches, the chunked dW, Adam on the owned shard)
rm -rf $O/stats_dp8; rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_dp8 -- python3 $R/bench.py --steps 60 --warmup 5 --dp-emulate 8 > $O/stats_dp8.log 2>&1
# PMC passes (each on its own, no tracing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
# doc2vec (round 5: PV-DM's word-vector additions deferred to one per kept position): passes, kernel stats, and the bytes through the fabric, with and without (NTF_D2V_DEFER=0)
python3 $R/profiles/d2v_pass.py > $O/d2v_passes.txt 2>&1
NTF_D2V_DEFER=0 python3 $R/profiles/d2v_pass.py > $O/d2v_passes_NTF_D2V_DEFER_0.txt 2>&1
rm -rf $O/d2v_stats $O/d2v_pmc_fetch $O/d2v_pmc_write $O/d2v_pmc_tcc
rocprofv3 --kernel-trace --stats --output-format csv -d $O/d2v_stats -- python3 $R/profiles/d2v_pass.py > $O/d2v_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/d2v_pmc_fetch -- python3 $R/profiles/d2v_pass.py > $O/d2v_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/d2v_pmc_write -- python3 $R/profiles/d2v_pass.py > $O/d2v_pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum --output-format csv -d $O/d2v_pmc_tcc -- python3 $R/profiles/d2v_pass.py > $O/d2v_pmc_tcc.log 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $O | tail -1
