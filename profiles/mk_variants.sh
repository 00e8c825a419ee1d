#!/bin/bash
# Diagnostic builds of the library beside the shipped one (never shipped; selected per process with NTF_LIB_PATH):
#   profiles/mk_variants.sh diag   -> scratch/var/diag.so       -DNTF_DIAG: ablations, stamps, NTF_SKIP, NTF_DW_TAIL, the co-scheduling experiment
#   profiles/mk_variants.sh ieee   -> scratch/var/adam_ieee.so  -DNTF_ADAM_IEEE: adam_step on sqrtf and a true division (r5_ep_tolerance.md)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); D=$R/scratch/var; mkdir -p $D/obj
cd $R/opentf_amd/csrc
W="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-unused-value"
case "$1" in
  diag) F="$W -DNTF_DIAG"; out=$D/diag.so ;;
  ieee) F="$W -DNTF_ADAM_IEEE"; out=$D/adam_ieee.so ;;
  def) F="$W $3"; out=$D/$2.so ;;      # profiles/mk_variants.sh def NAME "-DMACRO=value ..."  -> scratch/var/NAME.so (A/B of a compile-time choice)
  *) echo "usage: $0 diag|ieee|def NAME FLAGS"; exit 2 ;;
esac
for f in ntf_kernels ntf_head ntf_engine ntf_metrics ntf_cooc ntf_n2v ntf_d2v; do hipcc $F -c $f.hip -o $D/obj/$f.o 2>/dev/null & done
for f in ntf_fused ntf_special ntf_fused_dw; do hipcc $F -fno-slp-vectorize -c $f.hip -o $D/obj/$f.o 2>/dev/null & done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $out $D/obj/*.o
rm -rf $D/obj; echo built $out
