"""k_out_dw_q per-wave stamps (a -DNTF_DIAG build of the library, NTF_DW_STAMP_FILE=<file>): entry / main loop / exit times on the 100 MHz clock, the CU a workgroup ran
on, its LDS base (0 = the first workgroup of its CU) and shader-clock sums of the K blocks (body, DMA wait, barrier) -> how long main loop and epilogue take, and for
which share of a CU's time one workgroup was in its main loop while the other streamed its epilogue.

    python profiles/dw_stamps.py STAMPS.bin
"""
import numpy as np, sys
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4, 10)   # [workgroup][wave][6]
w0 = a[:, 0, :]
t0 = w0[:, 0].min()
ent = (w0[:, 0] - t0) / 100.0; mb = (w0[:, 1] - t0) / 100.0; me = (w0[:, 2] - t0) / 100.0; ex = (w0[:, 3] - t0) / 100.0   # us
hw = w0[:, 4] & 0xffffffff; xcc = w0[:, 4] >> 32; lds = w0[:, 5]
cu = ((xcc.astype(np.int64) & 15) << 8) | ((hw.astype(np.int64) >> 8) & 0xff)
print("workgroups", len(w0), "distinct CUs", len(np.unique(cu)), "kernel span us", ex.max())
print("lds_base values", np.unique(lds, return_counts=True))
print("first-round (first 512) lds_base != 0:", int((lds[:512] != 0).sum()))
print("main loop us: mean %.1f  p10 %.1f p90 %.1f | epilogue us: mean %.1f p10 %.1f p90 %.1f | stagger wait mean %.1f" % (
    (me - mb).mean(), np.percentile(me - mb, 10), np.percentile(me - mb, 90), (ex - me).mean(), np.percentile(ex - me, 10), np.percentile(ex - me, 90), (mb - ent).mean()))
tot_ov = 0.0; tot_main2 = 0.0; tot_epi2 = 0.0; tot = 0.0
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    ev = []
    for i in idx: ev += [(mb[i], 'm', 1), (me[i], 'm', -1), (me[i], 'e', 1), (ex[i], 'e', -1)]
    ev.sort()
    nm = ne = 0; last = ev[0][0]
    for t, k, d in ev:
        dt = t - last; last = t
        if nm >= 1 and ne >= 1: tot_ov += dt
        if nm >= 2: tot_main2 += dt
        if ne >= 2: tot_epi2 += dt
        tot += dt
        if k == 'm': nm += d
        else: ne += d
print("cycles per K block and wave (all waves): body %.0f | DMA wait %.0f | barrier %.0f" % tuple(a[:, :, 6 + k].mean() / 32 for k in range(3)))
print("per-CU time: main||epilogue %.1f %%  two main loops %.1f %%  two epilogues %.1f %%" % (100 * tot_ov / tot, 100 * tot_main2 / tot, 100 * tot_epi2 / tot))
c = np.unique(cu)[3]
idx = np.where(cu == c)[0]; idx = idx[np.argsort(ent[idx])]
print("CU", hex(c))
for i in idx: print("  wg %5d lds %3d enter %7.1f main %7.1f..%7.1f exit %7.1f" % (i, lds[i], ent[i], mb[i], me[i], ex[i]))
