#!/usr/bin/env python
"""The numbers that DESIGN.md section 5 and profiles/README.md quote, generated FROM the committed artefacts of a round
(the prose drifted from the CSVs it cited).

    python profiles/make_tables.py            # print the generated blocks
    python profiles/make_tables.py --write    # rewrite the marked blocks of DESIGN.md and profiles/README.md in place

Inputs (TAG = r06): <TAG>_kernel_stats.csv / <TAG>_kernel_stats_fft.csv (rocprofv3 --kernel-trace --stats of the bench
command), <TAG>_bench_under_rocprof[_fft].json (the lines those profiled runs printed: HIP-event averages),
<TAG>_bench.json (the default run), <TAG>_secondary_kernels.json, <TAG>_gputest.log, lds_issue_rates.json.
tests/test_docs_tables.py fails when the blocks in the two documents differ from what this script generates, and when a
kernel time of the profiled run differs from the CSV's by more than 2 %.
"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
TAG = "r06"
HBM_PEAK = 8.0e12
BEGIN = "<!-- BEGIN GENERATED %s (profiles/make_tables.py) -->"
END = "<!-- END GENERATED %s -->"


def last_json_line(path):
    for line in reversed(open(path).read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise ValueError("no JSON line in " + path)


def detail_line(path):
    for line in open(path).read().splitlines():
        if line.startswith("# detail: "):
            return json.loads(line[len("# detail: "):])
    return None


def kernel_stats(path):
    """{short kernel name: (calls, average ns)} of the library's kernels in a rocprofv3 stats CSV (instantiations of one
    template are kept apart by their template arguments)."""
    out = {}
    for r in csv.DictReader(open(path)):
        name = r["Name"]
        if "ffsa::" not in name:
            continue
        short = name.split("ffsa::", 1)[1].split("(", 1)[0]
        out[short] = (int(r["Calls"]), float(r["AverageNs"]))
    return out


def find(stats, prefix):
    """the most-called (then longest-running) instantiation whose name starts with `prefix`"""
    cands = [(v[0], v[1], k) for k, v in stats.items() if k == prefix or k.startswith(prefix + "<")]
    if not cands:
        return None, (0, float("nan"))
    k = max(cands)[2]  # (most calls, then the longest: the diagnostic instantiations are launched as often but exit at once)
    return k, stats[k]


def load():
    a = {}
    a["stats"] = kernel_stats(os.path.join(PROF, TAG + "_kernel_stats.csv"))
    a["stats_fft"] = kernel_stats(os.path.join(PROF, TAG + "_kernel_stats_fft.csv"))
    a["prof"] = last_json_line(os.path.join(PROF, TAG + "_bench_under_rocprof.json"))
    a["prof_fft"] = last_json_line(os.path.join(PROF, TAG + "_bench_under_rocprof_fft.json"))
    a["bench"] = last_json_line(os.path.join(PROF, TAG + "_bench.json"))
    a["detail"] = detail_line(os.path.join(PROF, TAG + "_bench.json")) or {}
    sec = os.path.join(PROF, TAG + "_secondary_kernels.json")
    a["secondary"] = json.load(open(sec)) if os.path.exists(sec) else {}
    a["lds"] = json.load(open(os.path.join(PROF, "lds_issue_rates.json")))
    log = os.path.join(PROF, TAG + "_gputest.log")
    a["gputest"] = open(log).read() if os.path.exists(log) else ""
    return a


def agreement(a):
    """[(kernel, rocprofv3 us per launch, HIP-event us per launch of the same profiled run)]"""
    rows = []
    pairs_per_launch = a["prof"]["config"]["pairs_per_gpu"]
    for key, kname in (("runs_corr", "k_runs_corr"), ("runs_extract", "k_runs_extract")):
        _, (_, avg_ns) = find(a["stats"], kname)
        rows.append((kname, avg_ns / 1e3, a["prof"]["kernels_us_per_pair"][key] * pairs_per_launch))
    fpl = a["prof_fft"]["config"]["pairs_in_flight"]
    for key, kname in (("mid", "k_mid_seg_one"), ("pass_a", "k_pass_a"), ("pass_c", "k_pass_c_pruned"), ("rescore", "k_rescore")):
        _, (_, avg_ns) = find(a["stats_fft"], kname)
        ev = a["prof_fft"]["kernels_us_per_pair"][key] * fpl
        if kname == "k_pass_c_pruned":  # (its exhaustive instantiation shares the HIP-event span)
            extra = [v[1] for k, v in a["stats_fft"].items() if k.startswith("k_pass_c_pruned<") and k.endswith("true>")]
            avg_ns += extra[0] if extra else 0.0
        rows.append((kname, avg_ns / 1e3, ev))
    return rows


def n_passed(a):
    m = re.search(r"(\d+) passed", a["gputest"])
    return int(m.group(1)) if m else None


def design_block(a):
    b, d = a["bench"], a["bench"].get("detail", {})
    rl, rh, rf = b["roofline"], b["roofline_hbm"], b["fft_path_roofline"]
    ppl = b["config"]["pairs_per_gpu"]
    _, (calls_c, corr_ns) = find(a["stats"], "k_runs_corr")
    _, (calls_e, ext_ns) = find(a["stats"], "k_runs_extract")
    _, (_, mid_ns) = find(a["stats_fft"], "k_mid_seg_one")
    lds = a["lds"]
    row = lds["ds_add_u32_32_waves_per_cu"]
    L = []
    L.append("**Round-6 numbers** (MI355X, one GPU, `python bench.py`: %d pairs x 7 ratios per step, inputs resident in HBM; every "
             "figure below is read from `profiles/%s_*` by `profiles/make_tables.py` -- `tests/test_docs_tables.py` fails when this "
             "block and the artefacts disagree)." % (ppl, TAG))
    L.append("")
    L.append("| | value | source |")
    L.append("|---|---|---|")
    L.append("| headline: seven-ratio solves/s from bits (auto -> run-boundary path) | **%.2f M** (%.3f ms per step) | `%s_bench.json` |"
             % (b["value"] / 1e6, b["ms_per_step"], TAG))
    L.append("| `k_runs_extract` | %.4f us/pair; must-move %.0f GB/s = **%.3f** of 8 TB/s; PMC traffic / must-move %.3f | `roofline_hbm` |"
             % (b["kernels_us_per_pair"]["runs_extract"], rh["achieved"], rh["frac"], rh.get("wasted") or float("nan")))
    L.append("| `k_runs_corr` | %.4f us/pair; %.0f G scatter-adds/s = **%.3f** of the random-word LDS ceiling (%.0f G/s: `ds_add_u32` "
             "%.2f cycles per wave-instruction), %.3f of the conflict-free rate (%.2f cycles) | `roofline`, `lds_issue_rates.json` |"
             % (b["kernels_us_per_pair"]["runs_corr"], rl["achieved"], rl["frac"], rl["peak"], row["random_words"],
                rl.get("frac_of_conflict_free_rate") or float("nan"), row["conflict_free"]))
    L.append("| rocprofv3 of the same command | `k_runs_corr` %.1f us, `k_runs_extract` %.1f us per %d-pair launch (%d launches) | `%s_kernel_stats.csv` |"
             % (corr_ns / 1e3, ext_ns / 1e3, a["prof"]["config"]["pairs_per_gpu"], calls_c, TAG))
    L.append("| vectors resident as boundary lists | %.2f M solves/s | `detail.resident_lists_value` |" % (d.get("resident_lists_value", float("nan")) / 1e6))
    L.append("| transform path, same pairs (`fft_path`) | %.1f k solves/s; `k_mid_seg_one` %.1f us per %d-pair launch = **%.3f** of 8 TB/s "
             "(PMC traffic / must-move %.3f) | `fft_path_value`, `%s_kernel_stats_fft.csv` |"
             % (b["fft_path_value"] / 1e3, mid_ns / 1e3, a["prof_fft"]["config"]["pairs_in_flight"], rf["frac"], rf.get("wasted") or float("nan"), TAG))
    L.append("| reference's own transform length / windowless | %.1f k / %.1f k solves/s (windowless goldens %s) | `reference_length_value`, `windowless_value` |"
             % (b["reference_length_value"] / 1e3, b["windowless_value"] / 1e3, b.get("windowless_golden")))
    L.append("| offsets vs the unmodified reference | %s pairs of the timed batch; ground truth %s / %s | `offset_match` |"
             % (b["offset_match"]["pairs_matching_reference_golden"], b["offset_match"]["pairs_matching_ground_truth"], b["offset_match"]["pairs"]))
    sp, spl = d.get("strong_proxy", {}), d.get("strong_proxy_lists", {})
    L.append("| one rank's share at 8 GPUs (128 pairs per step) | %.3f ms from bits (%.2f of the large-batch rate), %.3f ms with resident lists | `detail.strong_proxy*` |"
             % (sp.get("ms_per_step", float("nan")), sp.get("efficiency_vs_headline_batch", float("nan")), spl.get("ms_per_step", float("nan"))))
    L.append("| other legs | float inputs %.0f k (goldens %s), gss %.1f k files/s (%s), byte inputs %.0f k, single ratio %.1f M / %.0f k (none), "
             "ingest from bits %.0f k, from lists %.0f k / %.0f k (resident tables) / %.0f k (TrackSet per batch), drop-in %.3f ms (device rasters) / "
             "%.2f ms (host float64), VAD sweep %.0f GB/s = %.3f, end to end %.0f files/s | `detail` |"
             % (d.get("float_inputs", 0) / 1e3, d.get("float_inputs_golden"), d.get("gss_files_per_s", 0) / 1e3, d.get("gss_all_files_equal_per_file_search"),
                d.get("byte_inputs", 0) / 1e3, d.get("single_ratio_6000", 0) / 1e6, d.get("single_ratio_none", 0) / 1e3,
                d.get("ingest_warm_plan_stream_pairs_per_s", 0) / 1e3, d.get("ingest_lists_warm_pairs_per_s", 0) / 1e3,
                d.get("ingest_lists_resident_tracks_pairs_per_s", 0) / 1e3, d.get("ingest_lists_trackset_per_batch_pairs_per_s", 0) / 1e3,
                d.get("drop_in_ms_device_rasters", float("nan")), d.get("drop_in_ms_host_arrays", float("nan")), d.get("vad_GBps", 0), d.get("vad_frac", 0),
                d.get("end_to_end_files_per_s", 0)))
    cb = b.get("cpu_baseline", {})
    L.append("| CPU baseline (golden-pinned restatement, `kind` = %s) | %.2f solves/s on %s core, %.1f on %s | `cpu_baseline` |"
             % (cb.get("kind"), cb.get("value", float("nan")), cb.get("cores"), cb.get("parallel", {}).get("value", float("nan")), cb.get("parallel", {}).get("cores")))
    bd = d.get("boundary_density") or []
    if bd:
        L.append("| boundary density sweep (boundaries per vector: auto / fft solves/s, path) | %s | `detail.boundary_density` |"
                 % "; ".join("%d: %.0f k / %.1f k (%s)" % (r[0], r[1] / 1e3, r[2] / 1e3, "runs" if r[3] == "r" else "transforms") for r in bd))
    n = n_passed(a)
    if n is not None:
        L.append("| `pytest -m gpu` of the same binary | %d passed | `%s_gputest.log` |" % (n, TAG))
    L.append("")
    L.append("Agreement rocprofv3 <-> HIP events (us per launch, the profiled runs' own lines): "
             + ", ".join("`%s` %.1f vs %.1f" % r for r in agreement(a)) + ".")
    return "\n".join(L)


def readme_block(a):
    L = ["Agreement check (%s; us per launch, rocprofv3 `AverageNs` vs the HIP-event average of the same profiled run): " % TAG
         + "; ".join("`%s` %.1f vs %.1f" % r for r in agreement(a)) + "."]
    sec = a["secondary"].get("kernels", a["secondary"]) if isinstance(a["secondary"], dict) else {}
    picks = []
    for k in ("k_vad_energy[fp32 labels]", "k_vad_energy[bit-packed labels]", "k_vad_tokenize_scan", "k_speech_bounds", "k_rasterize_runs",
              "k_rasterize_batch", "k_runs_extract_lists", "k_levels_bits<double>", "k_pack_bits<0>"):
        v = sec.get(k) if isinstance(sec, dict) else None
        if isinstance(v, dict) and "avg_us" in v:
            picks.append("`%s` %.1f us (%.3f of 8 TB/s, PMC traffic / algorithmic bytes %.2f)"
                         % (k, v["avg_us"], v.get("frac_of_8TBps", float("nan")), v.get("pmc_over_algorithmic", float("nan"))))
    if picks:
        L.append("Secondary kernels (`%s_secondary_kernels.json`): %s." % (TAG, ", ".join(picks)))
    return "\n".join(L)


def replace_block(text, name, block):
    b, e = BEGIN % name, END % name
    if b not in text or e not in text:
        raise ValueError("markers for %s not found" % name)
    head, rest = text.split(b, 1)
    _, tail = rest.split(e, 1)
    return head + b + "\n" + block + "\n" + e + tail


def current_block(text, name):
    b, e = BEGIN % name, END % name
    if b not in text or e not in text:
        return None
    return text.split(b, 1)[1].split(e, 1)[0].strip("\n")


def main():
    a = load()
    blocks = {"design": design_block(a), "readme": readme_block(a)}
    if "--write" in sys.argv:
        for path, name in ((os.path.join(ROOT, "DESIGN.md"), "design"), (os.path.join(PROF, "README.md"), "readme")):
            text = open(path).read()
            open(path, "w").write(replace_block(text, name, blocks[name]))
        print("rewrote the generated blocks of DESIGN.md and profiles/README.md")
    else:
        for name, blk in blocks.items():
            print("=====", name)
            print(blk)


if __name__ == "__main__":
    main()
