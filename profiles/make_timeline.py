#!/usr/bin/env python3
"""rocprofv3 --kernel-trace csv -> the timeline of two consecutive train steps (both / all three streams) and a gap table.

    python profiles/make_timeline.py KERNEL_TRACE.csv OUT.csv OUT.md

OUT.csv: one row per dispatch of the two steps: kernel, queue (= HIP stream), start / end in microseconds from the first step's forward kernel, duration, gap to the
previous dispatch on the same queue.  OUT.md: the main stream's chain between the two big kernels with its gaps.  NOTE: under rocprofv3 every dependent launch boundary
reads 11-19 us (1.5-2 us un-profiled: MI355X_MICROARCH.md, row `boundary`) - the durations are the kernels' own, the gaps are the profiler's."""
import csv
import re
import sys

src, out_csv, out_md = sys.argv[1:4]
rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("ntf::", "")
    return n[:70]


names = [short(r["Kernel_Name"]) for r in rows]
fwd = [i for i, n in enumerate(names) if n.startswith("k_out_fwd_h3p") ]      # the TRAIN step's forward kernel (round 6: the run also ends with evaluation steps, k_out_fwd_h3e)
while len(fwd) >= 3 and any(n.startswith("k_out_fwd_h3e") for n in names[fwd[-3]: fwd[-1]]): fwd.pop()      # (two whole train steps with no evaluation step between them)
a, b = fwd[-3], fwd[-1]          # two whole steps: from one forward kernel to the one two steps later
t0 = int(rows[a]["Start_Timestamp"])
prev_end, out = {}, []
for i in range(a, b + 1):
    r = rows[i]
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    prev_end[q] = e
    out.append((names[i], q, (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, r["Grid_Size_X"], r["Workgroup_Size_X"]))
with open(out_csv, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "queue", "start_us", "end_us", "duration_us", "gap_to_previous_on_queue_us", "grid_x", "workgroup_x"])
    for o in out:
        w.writerow([o[0], o[1]] + [f"{x:.1f}" for x in o[2:6]] + list(o[6:]))
main_q = out[0][1]
step = (int(rows[fwd[-2]]["Start_Timestamp"]) - t0) / 1e3
with open(out_md, "w") as f:
    f.write("# One train step on the main stream (rocprofv3 --kernel-trace; defaults of the round named in the file name)\n\n")
    f.write(f"Step period in this PROFILED run: {step:.0f} us (un-profiled: see the round's bench_n1.json).  Gaps are inflated by the profiler (11-19 us per dependent launch; 1.5-2 us un-profiled).\n\n")
    f.write("| kernel | queue | start us | duration us | gap before us |\n|---|---|---|---|---|\n")
    for o in out:
        if o[2] > step + 1: break
        f.write(f"| `{o[0]}` | {o[1]} | {o[2]:.1f} | {o[4]:.1f} | {o[5]:.1f} |\n")
    ks = [o for o in out if o[2] < step - 1 and o[1] == main_q]        # one step: up to, not including, the next step's forward kernel (the table's closing row)
    is_big = lambda o: o[0].startswith(("k_out_fwd_h3", "k_out_dw_p2", "k_out_dw_q"))
    big = sum(o[4] for o in ks if is_big(o))
    small = sum(o[4] for o in ks) - big
    f.write(f"\nMain stream, one step: the two big kernels {big:.0f} us, every other kernel {small:.0f} us in {sum(1 for o in ks if not is_big(o))} launches; side streams run beside them.\n")
print("wrote", out_csv, out_md)
