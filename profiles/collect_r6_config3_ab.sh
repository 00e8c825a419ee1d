#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r6c3; mkdir -p $O; cd $R
python -m pytest tests/test_gpu_replay.py -x -q -m gpu -k "multihot or config2" 2>&1 | tail -8 > $O/t_replay.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -x -q -m gpu -k "multihot or config3" 2>&1 | tail -8 > $O/t_multihot.log
B="bench.py --no-cpu-baseline --no-extra-configs --no-f32-line --no-gather-bench --steps 30 --warmup 5 --input multihot --nsd unigram"
for i in 1 2; do
  NTF_L0_SWEEP=1 python $B > $O/c3_sweep1_$i.json 2>> $O/bench.err
  NTF_L0_SWEEP=0 python $B > $O/c3_sweep0_$i.json 2>> $O/bench.err
done
tail -3 $O/t_replay.log $O/t_multihot.log
