#!/usr/bin/env python
"""PMC HBM traffic per pair per kernel -> profiles/traffic_per_pair.json (read by bench.py).

    python profiles/make_traffic.py <pmc_dir> <n_fft_device> <pairs_per_launch>

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: rocprofv3 reports both in KB, and on
gfx950 FETCH_SIZE counts coalesced streaming reads at half their size (MI355X_MICROARCH.md, HBM
section).  The factor is calibrated here on k_mid and k_pass_c, whose read volumes are known
(80 resp. 64 bytes x n_fft per pair) -- the corrected figure matches them within ~5 %.
"""
import collections
import csv
import glob
import json
import os
import sys

pmc_dir, n_dev, ppl = sys.argv[1], sys.argv[2], float(sys.argv[3])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
sq = collections.defaultdict(lambda: collections.defaultdict(list))
# wave-instruction and LDS-array counters (summed over the device by rocprofv3), reported per pair next to the bytes
SQ_NAMES = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_IDX_ACTIVE",
            "SQ_LDS_BANK_CONFLICT", "SQ_WAVES")
for f in glob.glob(pmc_dir + "/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        kn = row["Kernel_Name"]
        # the exhaustive-collect instantiations (k_pass_c_pruned<.., true>, k_pass_c<.., 2>) exit at once
        # for unflagged transforms; they are separate launches and must not dilute the pass-C average
        if "k_pass_c_pruned<" in kn and ", true>" in kn:
            continue
        if "k_pass_c<" in kn and ", 2>" in kn:
            continue
        if "ffsa::k_" in kn and row["Counter_Name"] in SQ_NAMES:
            k = kn.split("ffsa::k_")[1].split("<")[0].split("(")[0]
            k = k.replace("pass_a3", "pass_a").replace("pass_c3", "pass_c").replace("pass_c_pruned", "pass_c").replace("mid_seg_one", "mid").replace("mid_seg_pipe", "mid").replace("mid_seg", "mid")
            sq[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        if "ffsa::k_" in kn and row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            k = row["Kernel_Name"].split("ffsa::k_")[1].split("<")[0].split("(")[0]
            k = k.replace("pass_a3", "pass_a").replace("pass_c3", "pass_c").replace("pass_c_pruned", "pass_c").replace("mid_seg_one", "mid").replace("mid_seg_pipe", "mid").replace("mid_seg", "mid")  # bench.py's kernel ids
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k, c in acc.items():
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"])
        write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"])
        # pass A reads bytes (narrow loads): the half-size artefact is documented for wide streaming
        # reads only, so its FETCH_SIZE is taken at face value (an upper-bound-free, uncalibrated figure)
        fmul = 1 if k == "pass_a" else 2
        out[k] = (fmul * fetch + write) * 1024 / ppl
        if k.startswith("runs_"):  # raw counters next to the corrected figure (k_runs_extract's read volume is known: the bits)
            out[k + "_raw_fetch_KB_write_KB_per_launch"] = [fetch, write]
work = {k: {c.replace("SQ_", "").lower() + "_per_pair": sum(v) / len(v) / ppl for c, v in cs.items()} for k, cs in sq.items()
        if k in ("pass_a", "mid", "pass_c", "nominees", "rescore", "runs_extract", "runs_corr")}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic_per_pair.json")
data = json.load(open(path)) if os.path.exists(path) else {}
data.setdefault(n_dev, {}).update(out)  # (the run-boundary and the transform kernels come from separate passes)
if work:
    data.setdefault(n_dev + "_work", {}).update(work)
json.dump(data, open(path, "w"), indent=1, sort_keys=True)
print(json.dumps({n_dev: out}, indent=1))
