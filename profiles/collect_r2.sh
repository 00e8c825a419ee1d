#!/bin/bash
# Round-2 profile collection on the GPU box:  /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect_r2.sh'
# Outputs land in gpurun_out/r2/ ; the summaries are folded into profiles/r2_* by profiles/fold_r2.sh here afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-f32-line"
# bench lines (un-profiled)
python3 $R/bench.py --steps 50 --warmup 10 > $O/bench_n1.json 2> $O/bench_n1.err
python3 $B --no-gather-bench --steps 50 --warmup 10 --model fnn > $O/bench_n1_fnn.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --mfma bf16x6 > $O/bench_n1_bf16x6.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --mfma f32 > $O/bench_n1_f32mfma.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 20 --warmup 3 --dataset dblp_full --rows 200000 > $O/bench_n1_dblp_full.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --input multihot --nsd unigram > $O/bench_n1_config3_multihot_unigram.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset uspt --d 256 > $O/bench_n1_config4_uspt_d256.json 2>> $O/bench.err
python3 $B --no-gather-bench --steps 30 --warmup 5 --dataset gith > $O/bench_n1_config5_gith.json 2>> $O/bench.err
# expert-sharded multi-GPU path: what ONE rank of G runs (its 1/G of the experts, G x 1000 rows, two-phase step, no exchange), emulated
# on this GPU, and the sharded step through RCCL at world size 1
for G in 2 4 8; do python3 $R/bench.py --steps 20 --warmup 4 --ep-emulate $G > $O/bench_ep_rank_of_$G.json 2>> $O/bench.err; done
python3 $B --no-gather-bench --steps 20 --warmup 4 --parallel ep --force-dist > $O/bench_ep_world1_rccl.json 2>> $O/bench.err
# kernel trace + stats of the default run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --steps 20 --warmup 3 > $O/stats.log 2>&1
# PMC passes (each on its own, no tracing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
# the whole-dataset gather launch on its own
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_gather_fetch -- python3 $B --gather-only > $O/pmc_gather_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_gather_write -- python3 $B --gather-only > $O/pmc_gather_write.log 2>&1
# keep what is small: stats csv + counter csvs
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $O | tail -1
