"""GPU timeline of small steps from a rocprofv3 kernel trace: python profiles/small_step_timeline.py <kernel_trace.csv>
prints, for the last few steps of profiles/small_step_profile.py (bits), every kernel with its start relative to the step's first
kernel, its duration and the idle gap in front of it."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].replace("ffsa::", "").replace("void ", "")[:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# steps start at k_runs_extract*
idx = [i for i, k in enumerate(ks) if k[0].startswith("k_runs_extract")]
for a, b in zip(idx[-25:-22], idx[-24:-21]):
    t0 = ks[a][1]
    prev_end = ks[a - 1][2] if a else t0
    print("--- step (previous kernel ended %.1f us before)" % ((t0 - prev_end) / 1e3))
    for name, s, e in ks[a:b]:
        print("  %-40s start %7.1f us  dur %6.1f us  gap %5.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
