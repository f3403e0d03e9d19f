import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(r["Kernel_Name"].split("(")[0].replace("ffsa::", "").replace("void ", "")[:40], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
idx = [i for i, k in enumerate(ks) if k[0].startswith("k_runs_extract")]
for a, b in zip(idx[-4:-1], idx[-3:]):
    t0 = ks[a][1]
    prev_end = ks[a - 1][2] if a else t0
    print("--- step (previous kernel ended %.1f us before)" % ((t0 - prev_end) / 1e3))
    for name, s, e in ks[a:b]:
        print("  %-40s start %7.1f us  dur %6.1f us  gap %5.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
        prev_end = e
