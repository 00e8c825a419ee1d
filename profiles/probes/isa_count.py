#!/usr/bin/env python3
"""Count instructions by issue class between two line numbers of a hipcc -save-temps .s listing, and price them with MI355X_MICROARCH.md's
'vector-instruction ISSUE cost' constants (one wave's stream on one SIMD).  usage: isa_count.py file.s first_line last_line [label]"""
import re, sys, collections
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
def klass(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith(TRANS): return "valu_trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"): return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"): return "lds_write"
    if op.startswith("global_load_lds") or (op.startswith("buffer_load") and False): return "lds_dma"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_"): return "salu"
    return "other"
COST = {"mfma": 8, "valu_trans": 8, "valu": 4, "lds_read": 4, "lds_write": 4, "lds_dma": 60, "vmem": 4, "s_nop": 4, "salu": 0, "waitcnt": 0, "barrier": 0, "other": 0}
def main():
    f, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    label = sys.argv[4] if len(sys.argv) > 4 else ""
    cnt = collections.Counter(); ops = collections.Counter()
    for ln, line in enumerate(open(f), 1):
        if ln < a or ln > b: continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        if "lds" in t and op.startswith("global_load") : k = "lds_dma"
        else: k = klass(op)
        cnt[k] += 1; ops[op] += 1
    tot = sum(cnt[k] * COST[k] for k in cnt)
    print(f"## {label} lines {a}-{b}")
    for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]): print(f"  {k:11s} {v:5d}  x{COST[k]:3d} = {v * COST[k]:6d}")
    print(f"  issue cycles (priced): {tot}")
    print("  top ops: " + ", ".join(f"{o} {n}" for o, n in ops.most_common(28)))
if __name__ == "__main__": main()
