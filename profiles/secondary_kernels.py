#!/usr/bin/env python
"""Every kernel the bench line quotes OUTSIDE the timed headline, a few launches each, so that ONE
`rocprofv3 --kernel-trace --stats` (and one `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass) covers them:

    k_vad_energy (fp32 labels / bit-packed labels), k_vad_tokenize_scan, k_speech_bounds, k_pack_bits,
    k_rasterize_batch, k_levels_bits (multi-level float references), and the transform kernels of the windowless (3*2^19: k_pass_a3, k_mid, k_pass_c3) and
    reference-length (2^21) plans plus the window-shortened default (k_pass_a, k_mid_seg_one, k_pass_c_pruned).

Prints ONE JSON line: the algorithmic bytes per launch of each kernel (what profiles/summarize_secondary.py divides the
rocprofv3 durations into).  python profiles/secondary_kernels.py [pairs]      (GPU box)
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ffsubsync_amd import _native, batch  # noqa: E402
from ffsubsync_amd.constants import candidate_ratios  # noqa: E402
from workloads import synth  # noqa: E402

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 512
REP = 4
alg = {}

# ---- VAD sweep: one 90-minute 48 kHz s16le file (BASELINE configs[4] shape), as bench.py's vad leg builds it
frame, n_frames = 480, 90 * 60 * 100
n = n_frames * frame
g = torch.Generator(device="cuda")
g.manual_seed(1234)
seg = torch.randint(0, 2, (n_frames // 50 + 1,), generator=g, device="cuda").repeat_interleave(50)[:n_frames].bool()
sigma = torch.where(seg, 3000.0, 30.0).repeat_interleave(frame)
pcm = (torch.randn(n, generator=g, device="cuda") * sigma).round().clamp(-32768, 32767).to(torch.int16)
del sigma
words = torch.zeros((n_frames + 31) // 32, dtype=torch.int32, device="cuda")
for _ in range(REP):
    labels = _native.vad_energy(pcm, frame, 50.0, 0.0)
for _ in range(REP):
    _native.vad_energy_bits(pcm, frame, 50.0, out=words)
# the two entry points share k_vad_energy: launches alternate fp32 / bits in the trace (REP each, in this order)
alg["k_vad_energy"] = {"bytes_per_launch_fp32_labels": 2 * n + 4 * n_frames, "bytes_per_launch_bit_labels": 2 * n + n_frames // 8,
                       "launch_order": "first %d launches write fp32 labels, next %d bit-packed labels" % (REP, REP)}
for _ in range(REP):
    _native.vad_tokenize(labels, 10000, 20, 500, 25, 0.0)
alg["k_vad_tokenize_scan"] = {"bytes_per_launch": 8 * n_frames}  # reads and writes one fp32 label per frame
for _ in range(REP):
    _native.speech_bounds(labels)
alg["k_speech_bounds"] = {"bytes_per_launch": 4 * n_frames}
for _ in range(REP):
    _native.pack_bits(labels)
# (keyed by instantiation: k_pack_bits<1> packs fp32 labels here; k_pack_bits<0> packs the 0/1 BYTE chunks of
# synth.build_device_batch further down -- round 4's summary divided those 184 MB launches by this 2 MB figure: "93 x")
alg["k_pack_bits<1>"] = {"bytes_per_launch": 4 * n_frames + n_frames // 8}
del pcm, labels
torch.cuda.synchronize()

# ---- batched rasteriser: reference track + subtitle track at the seven ratios for `pairs` files, one call
specs = [synth.make_pair_spec(s) for s in range(pairs)]
recs = [((sp.ref_starts * 10000, sp.ref_ends * 10000, None), (sp.cand_starts[0] * 10000, sp.cand_ends[0] * 10000, None))
        for sp in specs]  # (reference track, candidate track), microseconds
try:
    for _ in range(REP):
        dbr = batch.pairs_from_intervals(recs, candidate_ratios())
    n_sub = sum(len(r[0][0]) + 7 * len(r[1][0]) for r in recs)
    alg["k_rasterize_batch"] = {"bytes_per_launch": int(dbr.data.numel()) + 16 * n_sub,
                                "what": "output bits (zeroed + written) + 16 bytes per (vector, subtitle) interval read"}
    del dbr
    # round 5: the same vectors as boundary lists -- no bitmap (subtitle tables uploaded once, as the searches hold them)
    ts = batch.TrackSet([t for rec in recs for t in rec]).to_device()
    track_of = (np.tile(np.array([0] + [1] * 7), (pairs, 1)) + 2 * np.arange(pairs)[:, None]).ravel()
    ratio = np.tile(np.array([1.0] + list(candidate_ratios())), (pairs, 1)).ravel()
    for _ in range(REP):
        data_l, offs_l, lens_l, bounds_l = ts.rasterize_runs(track_of, ratio)
    torch.cuda.synchronize()
    n_ent = int(sum(int(data_l[int(o): int(o) + 4].view(torch.int32).item()) + 1 for o in offs_l[:: max(1, len(offs_l) // 64)]) * max(1, len(offs_l) // 64))
    alg["k_rasterize_runs"] = {"bytes_per_launch": 16 * n_sub + 8 * n_ent + 16 * len(offs_l),
                               "what": "16 bytes per (vector, subtitle) interval read + 8 bytes per list entry and 16 per header "
                                       "written (entries counted on every %d-th vector)" % max(1, len(offs_l) // 64)}
    del data_l
except Exception as exc:  # the record layout of pairs_from_intervals is checked by its own tests; never take the trace down
    alg.setdefault("k_rasterize_batch", {"error": repr(exc)[:200]})
    alg.setdefault("k_rasterize_runs", {"error": repr(exc)[:200]})

# ---- transform kernels of the three plans (FFS_ALGO_FFT), `pairs` pairs per launch
db = synth.build_device_batch(specs)
alg["k_pack_bits<0>"] = {"bytes_per_launch": float(np.sum(db.lens)) * (1.0 + 1.0 / 8.0) / max(1, (pairs + 31) // 32),
                         "what": "synth.build_device_batch packing its 0/1 byte chunks of 32 pairs (bytes read + bits written)"}
# bits -> boundary lists for every vector of the batch (ffs_runs_from_bits_batch = k_runs_extract_lists)
for _ in range(2):
    dl = db.to_runs(cap=8192)
torch.cuda.synchronize()
n_b = dl.data.view(torch.int32).reshape(-1, int(dl.offs.ravel()[1]) // 4)[:, 0].sum().item()
alg["k_runs_extract_lists"] = {"bytes_per_launch": float(np.sum((db.lens + 31) // 32 * 4)) + 8.0 * n_b + 16.0 * db.lens.size,
                               "what": "every bit-packed vector read once + 8 bytes per boundary written"}
del dl
# ---- multi-level float64 references (the `weighted` fused VAD's four levels) against the bit-packed candidates: the level
# kernels (k_levels_sample, k_levels_bits: the float samples are read once), k_runs_extract on the planes + candidates, k_runs_corr_ml
try:
    n_f = min(pairs, 128)
    dbf = synth.build_fused_batch(specs[:n_f])
    alf = batch.BatchAligner(dbf.required_fft_length(6000), 7, 6000, pairs_in_flight=n_f)
    for _ in range(REP):
        alf.solve_async(dbf, 0, n_f)
    torch.cuda.synchronize()
    alf.close()
    ref_lens = dbf.lens[:, 0].astype(np.float64)
    alg["k_levels_bits"] = {"bytes_per_launch": float(np.sum(8.0 * ref_lens + 3.0 * np.ceil(ref_lens / 32.0) * 4.0)),
                            "what": "%d float64 references read once + three bit planes written each" % n_f}
    del dbf
except Exception as exc:
    alg["k_levels_bits"] = {"error": repr(exc)[:200]}
unit = lambda nfft: 8.0 * nfft
in_bytes = float(np.mean(db.lens.sum(axis=1))) / 8.0
for label, mo, ref_len in (("default_3x2^18_segmented", 6000, False), ("windowless_3x2^19", None, False), ("reference_length_2^21", 6000, True)):
    n_fft = db.required_fft_length(mo, reference_length=ref_len)
    al = batch.BatchAligner(n_fft, 7, mo, pairs_in_flight=pairs, algorithm="fft")
    for _ in range(2):
        al.solve_async(db, 0, pairs)
    torch.cuda.synchronize()
    al.close()
    seg_mode = label.startswith("default")
    s = 3.5
    mm = ({"pass_a": (s + 0.5) * unit(n_fft) + in_bytes, "mid": (s + 0.5 + s / 3.0) * unit(n_fft), "pass_c": s / 3.0 * unit(n_fft)}
          if seg_mode else
          {"pass_a": (s + 0.5) * unit(n_fft) + in_bytes, "mid": (2 * s + 0.5) * unit(n_fft), "pass_c": s * unit(n_fft)})
    alg["transforms_" + label] = {"n_fft": int(n_fft), "pairs_per_launch": pairs,
                                  "bytes_per_launch": {k: v * pairs for k, v in mm.items()}}
print(json.dumps(alg))
