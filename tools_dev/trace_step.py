#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace CSV: every kernel between two
consecutive clip_adam launches (late in the run), with its queue, start offset, duration, and
how much of it overlapped the previous kernel.  usage: trace_step.py <dir-or-csv> [step_from_end]"""
import csv, glob, sys
src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
f = src if src.endswith(".csv") else (glob.glob(src + "/*/*_kernel_trace.csv") + glob.glob(src + "/*_kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "clip_adam" in r["Kernel_Name"]]
if len(adam) < back + 2:
    sys.exit("not enough steps in the trace")
lo, hi = adam[-back - 1], adam[-back]
t0 = int(rows[lo]["End_Timestamp"])
print("step window: %.1f us (end of clip_adam to end of the next clip_adam)" % ((int(rows[hi]["End_Timestamp"]) - t0) / 1e3))
prev_end = t0
busy = 0
for r in rows[lo + 1:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("blh::", "").replace("void ", "").split("(")[0]
    for a, b in (("gemm_f32_ring_kernel", "ring"), ("_kernel", "")):
        name = name.replace(a, b)
    gap = (s - prev_end) / 1e3
    print("q%-2s +%8.1f us  dur %7.1f us  gap-after-prev-end %7.1f  %s  grid %sx%sx%s" % (
        r["Queue_Id"], (s - t0) / 1e3, (e - s) / 1e3, gap, name[:70],
        int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), r["Grid_Size_Y"], r["Grid_Size_Z"]))
    prev_end = max(prev_end, e)
