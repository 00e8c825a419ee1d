#!/bin/bash
# Round-3 profile collection on the GPU box:  /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash profiles/collect_r3.sh'
# Outputs land in gpurun_out/r3/ ; the summaries are folded into profiles/r3_* by profiles/fold_r3.sh here afterwards.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3
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
# A/B of this round's changes, same box (each env switch restores the round-2 behaviour of one piece)
for v in "NTF_FWD_KERNEL=1" "NTF_FWD_KERNEL=4" "NTF_PREFETCH=0" "NTF_HEAD=0" "NTF_SIDE_BWD=0"; do
  env $v python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/ab_$v.json 2>> $O/bench.err
done
# expert-sharded multi-GPU path: what ONE rank of G runs, emulated on this GPU
for G in 2 4 8; do python3 $R/bench.py --steps 20 --warmup 4 --ep-emulate $G --no-extra-configs > $O/bench_ep_rank_of_$G.json 2>> $O/bench.err; done
# the forward kernel's phases (in-kernel s_memtime stamps, diagnostic build of the same kernel)
NTF_FWD_ABL=9 python3 $B --no-gather-bench --steps 40 --warmup 10 > /dev/null 2> $O/fwd_stamps.err; grep "fwd stamps" $O/fwd_stamps.err > $O/fwd_stamps.txt
# package power and shader clock while the step runs (rocm-smi every 2 s beside a 20 000-step run), and the forward kernel's own clock (clock64 against wall_clock64)
python3 $B --no-gather-bench --steps 20000 --warmup 10 > $O/bench_long.json 2>> $O/bench.err &
BP=$!
rocm-smi --showmaxpower 2>/dev/null | grep "GPU\[" > $O/power_clocks.txt
for n in $(seq 1 40); do
  if ! kill -0 $BP 2>/dev/null; then break; fi
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | sed 's/.*: //' | tr '\n' ' ' >> $O/power_clocks.txt; echo >> $O/power_clocks.txt
  sleep 2
done
wait $BP
NTF_FWD_REPEAT=10 NTF_FWD_ABL=9 python3 $B --no-gather-bench --steps 40 --warmup 10 > /dev/null 2> $O/fwd_stamps_repeat10.err; grep "fwd stamps" $O/fwd_stamps_repeat10.err > $O/fwd_stamps_repeat10.txt
# kernel trace (every dispatch: the step timeline) + stats of the default run
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $B --no-gather-bench --steps 20 --warmup 3 > $O/stats.log 2>&1
# PMC passes (each on its own, no tracing)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_sq.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_lds -- python3 $B --no-gather-bench --steps 3 --warmup 1 > $O/pmc_lds.log 2>&1
find $O -name "*.db" -delete 2>/dev/null; find $O -name "*_agent_info.csv" -delete 2>/dev/null
du -sh $O | tail -1
