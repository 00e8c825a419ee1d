#!/bin/bash
# The -DNTF_DIAG part of the round-4 collection:  /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash profiles/collect_r4_diag.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-f32-line --no-extra-configs"
# -DNTF_DIAG build of the library (ablation / stamp / co-scheduling switches): the dW kernel's per-wave stamps, the forward kernel's phase stamps and clock
if [ -f $R/scratch/var/diag.so ] && nm -D $R/scratch/var/diag.so | grep -q ntf_head_prefetch_hits && strings $R/scratch/var/diag.so | grep -q "pair stamps"; then D=$R/scratch/var/diag.so; else
  D=/tmp/diag.so; cd $R/opentf_amd/csrc
  F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value -DNTF_DIAG"
  hipcc $F -fno-slp-vectorize -c ntf_fused.hip -o /tmp/diag_fused.o 2>/dev/null; hipcc $F -fno-slp-vectorize -c ntf_fused_dw.hip -o /tmp/diag_fused_dw.o 2>/dev/null; hipcc $F -c ntf_engine.hip -o /tmp/diag_engine.o 2>/dev/null
  hipcc --offload-arch=gfx950 -shared -fPIC -o $D ntf_kernels.o /tmp/diag_fused.o /tmp/diag_fused_dw.o ntf_head.o /tmp/diag_engine.o ntf_metrics.o ntf_cooc.o ntf_n2v.o ntf_d2v.o; cd /tmp
fi
NTF_LIB_PATH=$D NTF_DW_STAMP_FILE=$O/dw_stamps.bin python3 $B --no-gather-bench --steps 40 --warmup 10 > /dev/null 2>> $O/bench.err
python3 $R/profiles/dw_stamps.py $O/dw_stamps.bin > $O/dw_stamps.txt 2>&1; rm -f $O/dw_stamps.bin
# forward kernels: the wave-pair kernel's segment stamps (default), the one-wave kernel's phase stamps (NTF_FWD_KERNEL=3), and the one-wave kernel's ablations
NTF_LIB_PATH=$D NTF_FWD_ABL=9 python3 $B --no-gather-bench --steps 40 --warmup 10 > /dev/null 2> $O/fwd_pair_stamps.err; grep "pair stamps" $O/fwd_pair_stamps.err | head -3 > $O/fwd_pair_stamps.txt
NTF_LIB_PATH=$D NTF_FWD_KERNEL=3 NTF_FWD_ABL=9 python3 $B --no-gather-bench --steps 40 --warmup 10 > /dev/null 2> $O/fwd_stamps.err; grep "fwd stamps" $O/fwd_stamps.err > $O/fwd_stamps.txt
for a in 0 2 3; do NTF_LIB_PATH=$D NTF_FWD_KERNEL=3 NTF_FWD_ABL=$a python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/fwd_h3x_abl_$a.json 2>> $O/bench.err; done
if [ -n "$R4_COSCHED" ]; then
# co-scheduling experiment (NTF_COSCHED: RESULTS OF THESE RUNS ARE GARBAGE, timing only): the one-wave forward kernel on 8 x n workgroups alone, and beside the dW kernel
NTF_LIB_PATH=$D NTF_FWD_KERNEL=3 python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/cosched_serial.json 2>> $O/bench.err
for n in 32 28 24 20 16 12; do
  NTF_LIB_PATH=$D NTF_FWD_KERNEL=3 NTF_COSCHED=$n NTF_COSCHED_FWD_ONLY=1 python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/cosched_fwd_alone_$n.json 2>> $O/bench.err
  NTF_LIB_PATH=$D NTF_FWD_KERNEL=3 NTF_COSCHED=$n python3 $B --no-gather-bench --steps 40 --warmup 10 > $O/cosched_beside_dw_$n.json 2>> $O/bench.err
done
fi
