#!/usr/bin/env python3
"""Fold rocprofv3 --pmc csv outputs (one directory per pass) into one JSON: per kernel, per counter, the mean over dispatches, plus the
HBM traffic corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE tallies the 128-B requests
of wide coalesced reads at 64 B, so read bytes = 2 * FETCH_SIZE * 1024).

    python profiles/make_pmc_json.py OUT.json NOTE PASS_DIR [PASS_DIR ...]
"""
import collections
import csv
import glob
import json
import sys

out_path, note, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
kernels = {}
for k, d in acc.items():
    if not (k.startswith(("void ntf::", "ntf::")) or "k_d2v_" in k or "k_n2v_" in k):     # (the d2v / n2v kernels live in anonymous namespaces)
        continue
    c = {n: sum(v) / len(v) for n, v in d.items()}
    e = {"dispatches": max(len(v) for v in d.values()), "sq": {n: v for n, v in c.items() if n.startswith("SQ_")}}
    if "FETCH_SIZE" in c:
        e["FETCH_SIZE_KiB"] = c["FETCH_SIZE"]; e["hbm_read_bytes_corrected"] = 2 * c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c:
        e["WRITE_SIZE_KiB"] = c["WRITE_SIZE"]; e["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e["hbm_bytes"] = e["hbm_read_bytes_corrected"] + e["hbm_write_bytes"]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        e["mfma_busy_fraction_of_wave_cycles"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * c["SQ_WAVE_CYCLES"])   # WAVE_CYCLES counts quad-cycles
    kernels[k] = e
json.dump({"_note": note, "kernels": kernels}, open(out_path, "w"), indent=1)
print("wrote", out_path, len(kernels), "kernels")
