#!/usr/bin/env python
"""Headline benchmark: seven-ratio MaxScoreAligner(FFTAligner) solves per second on 2 h @ 100 Hz
activity vectors (BASELINE.json metric; workload = configs[2], 1024 pairs x 7 framerate ratios per
GPU, N = 2^21, max_offset_samples = 6000).

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (ffs_align_batch: pass A -> mid -> pass C -> nominees ->
exact re-evaluation -> max over ratios) over this rank's batch, inputs already resident in HBM,
plus the all-gather of the 24-byte per-pair results when N > 1.  Pairs are sharded by rank with
no data exchange during solves (weak scaling: every rank owns --pairs problems).  Prints ONE JSON
line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12  # B/s, MI355X spec (MI355X_MICROARCH.md)
HBM_COPY_CEILING = 6.29e12  # measured float4-copy ceiling, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1024, help="problems per GPU per step")
    ap.add_argument("--duration", type=float, default=7200.0, help="seconds of activity per vector")
    ap.add_argument("--pairs-in-flight", type=int, default=512)
    ap.add_argument("--cpu-pairs", type=int, default=12, help="pairs timed on the CPU oracle (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel HIP-event timing")
    ap.add_argument("--no-vad", action="store_true", help="skip the VAD frame-energy sweep figures")
    ap.add_argument("--e2e-files", type=int, default=8,
                    help="files in the end-to-end PCM -> VAD -> rasterise -> align figure (0 = skip)")
    ap.add_argument("--skip-full-length-record", action="store_true",
                    help="do not append the secondary full-length measurement (used by the PMC runs)")
    ap.add_argument("--full-length", action="store_true",
                    help="force the reference's transform length N=2^ceil(log2(R+S)) instead of the shorter "
                         "alias-free length the lag window allows")
    return ap.parse_args()


def vad_figures(torch, _native, minutes=90.0, iters=20):
    """Second kernel of the hot path: frame-energy VAD sweep over one 90-minute 48 kHz s16le file
    resident in HBM (BASELINE config 5 shape: 259.2 M samples, 518 MB).  Algorithmic bytes =
    2*n_samples + 4*n_frames."""
    from oracle import vad_oracle as vo

    frame = 480
    n_frames = int(minutes * 60 * 100)
    n = n_frames * frame
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    seg = torch.randint(0, 2, (n_frames // 50 + 1,), generator=g, device="cuda").repeat_interleave(50)[:n_frames].bool()
    sigma = torch.where(seg, 3000.0, 30.0).repeat_interleave(frame)
    pcm = (torch.randn(n, generator=g, device="cuda") * sigma).round().clamp(-32768, 32767).to(torch.int16)
    del sigma
    labels = _native.vad_energy(pcm, frame, 50.0, 0.0)
    ok = bool(torch.equal(labels > 0.5, seg))
    # bit-exact vs the CPU oracle on the first 100 s chunk (the reference's buffer size)
    head = pcm[: frame * 10000].cpu().numpy()
    ok_oracle = bool((labels[:10000].cpu().numpy() == vo.detect_fast(head)).all())
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    _native.vad_energy(pcm, frame, 50.0, 0.0)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(iters):
        _native.vad_energy(pcm, frame, 50.0, 0.0)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / iters
    bytes_ = 2 * n + 4 * n_frames
    t1 = time.perf_counter()
    cpu_n = frame * 10000 * 3
    vo.chunked_detect(pcm[:cpu_n].cpu().numpy())
    cpu_s = time.perf_counter() - t1
    lo, hi = _native.speech_bounds(labels)
    return {
        "workload": "one %.0f-min 48 kHz s16le file in HBM (%d samples, %d frames)" % (minutes, n, n_frames),
        "kernel": "k_vad_energy",
        "ms_per_file": ms,
        "audio_hours_per_s": minutes / 60.0 / (ms * 1e-3),
        "achieved_GBps": bytes_ / (ms * 1e-3) / 1e9,
        "frac_of_8TBps": bytes_ / (ms * 1e-3) / HBM_PEAK,
        "labels_match_generator": ok,
        "labels_match_cpu_oracle_first_chunk": ok_oracle,
        "speech_bounds": [lo, hi],
        "cpu_oracle_audio_hours_per_s": (cpu_n / 48000.0 / 3600.0) / cpu_s,
    }


def e2e_figures(torch, _native, n_files, minutes=90.0):
    """BASELINE config 5 at single-GPU scale: per file, 48 kHz s16le PCM resident in HBM -> frame-energy
    VAD -> 100 Hz activity vector; its subtitle file -> seven rasterised framerate-ratio candidates
    (from interval lists); then one batched MaxScoreAligner solve over all files.  Ground truth:
    every file's subtitles were stretched by one of the seven ratios and shifted by a known offset."""
    import numpy as np

    from ffsubsync_amd import batch, synth
    from ffsubsync_amd.constants import candidate_ratios
    from ffsubsync_amd.subtitle_raster import DeviceRaster, rasterize_candidates

    ratios = candidate_ratios()
    frame, n_frames = 480, int(minutes * 60 * 100)
    g = torch.Generator(device="cuda")
    g.manual_seed(99)
    files, truth = [], []
    for f in range(n_files):
        rng = np.random.RandomState(7000 + f)
        s_us, e_us, meta = synth.make_subtitle_records(7000 + f, duration_s=minutes * 60 * 0.9)
        idx, shift = int(rng.randint(7)), int(rng.randint(-4000, 4000))
        act = _native.rasterize_subtitles(s_us, e_us, meta, ratios[idx], 100, 0)  # where people speak
        speech = torch.zeros(n_frames, dtype=torch.bool, device="cuda")
        lo, hi = max(shift, 0), min(n_frames, shift + act.numel())
        speech[lo:hi] = act[lo - shift: hi - shift] != 0
        sigma = torch.where(speech, 3000.0, 30.0).repeat_interleave(frame)
        pcm = (torch.randn(n_frames * frame, generator=g, device="cuda") * sigma).round().clamp(-32768, 32767).to(torch.int16)
        del sigma
        files.append((pcm, (s_us, e_us, meta)))
        truth.append((idx, shift))
    torch.cuda.synchronize()

    def run():
        pairs = []
        for pcm, (s_us, e_us, meta) in files:
            labels = _native.vad_energy(pcm, frame, 50.0, 0.0)
            ref = DeviceRaster((labels > 0.5).to(torch.uint8), 0.0, 1.0)
            pairs.append((ref, rasterize_candidates(s_us, e_us, meta, ratios)))
        db = batch.pack_pairs(pairs)
        al = batch.BatchAligner(db.required_fft_length(6000), 7, 6000, pairs_in_flight=min(128, n_files))
        _, pres = al.solve(db)
        al.plan.close()
        return pres

    run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pres = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = sum(int(pres[i]["best_cand"]) == truth[i][0] and abs(int(pres[i]["offset"]) - truth[i][1]) <= 2
             for i in range(n_files))
    pcm_bytes = 2 * n_frames * frame
    return {
        "workload": "%d files x %.0f min: 48 kHz PCM (in HBM) -> VAD -> 7 rasterised candidates -> batched solve" % (n_files, minutes),
        "files_per_s_pcm_resident": n_files / dt,
        "ms_per_file_pcm_resident": 1e3 * dt / n_files,
        "recovered_ratio_and_offset": "%d/%d" % (ok, n_files),
        "pcie_bound_files_per_s_estimate": 63e9 / pcm_bytes,
        "note": "host-side per-file work here is the interval arithmetic of the rasteriser and Python launch overhead; "
                "streaming the PCM over PCIe Gen5 (63 GB/s spec) would cap one GPU at the estimate above",
    }


def main():
    args = parse()
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    dist = None
    force_dist = os.environ.get("FFS_BENCH_FORCE_DIST") == "1"  # exercise the RCCL path with one rank
    if world > 1 or force_dist:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from ffsubsync_amd import _native, batch, synth

    n_cand = 7
    P = args.pairs
    specs = [synth.make_pair_spec(rank * P + i, duration_s=args.duration) for i in range(P)]
    db = batch.build_device_batch(specs)
    n_ref = db.required_fft_length(None)  # the reference's N = 2^ceil(log2(R+S)) (aligners.py:67-68)
    # the +-6000 lag window lets the device use the shortest alias-free transform (ffs_plan_length)
    n_dev = n_ref if args.full_length else db.required_fft_length(6000)
    cand_out = torch.empty(P * n_cand * 24, dtype=torch.uint8, device="cuda")
    pair_out = torch.empty(P * 24, dtype=torch.uint8, device="cuda")
    use_dist = world > 1 or force_dist
    gathered = torch.empty(world * P * 24, dtype=torch.uint8, device="cuda") if use_dist else None
    profile = not args.no_profile

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_fft, steps, warmup):
        """W untimed + K timed passes over this rank's batch with a plan of length n_fft."""
        aligner = batch.BatchAligner(n_fft, n_cand, max_offset_samples=6000, pairs_in_flight=args.pairs_in_flight)

        def step():
            aligner.solve_async(db, 0, P, cand_out, pair_out)
            if use_dist:
                dist.all_gather_into_tensor(gathered, pair_out)

        for _ in range(warmup):
            step()
        aligner.plan.profile(profile)
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        ktimes = aligner.plan.profile_read() if profile else {}
        aligner.plan.profile(False)
        if use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        aligner.plan.close()
        return elapsed, ktimes

    # 168*N algorithmic bytes per seven-ratio solve (N = the reference's transform length)
    # = 42 half-transforms of 4*N bytes (each length-N fp32 transform = 8*N over two HBM passes):
    # pass A = 14 halves, mid = 21, pass C = 7.
    share = {"pass_a": 56, "mid": 84, "pass_c": 28}
    # bytes the kernels actually have to move per pair at device length n:
    # pass A writes 4 candidate transforms + the lower half of the reference's rows, mid reads those
    # 4.5 + writes 4, pass C reads 4 (8n bytes each)
    executed = {"pass_a": 4.5 * 8, "mid": 8.5 * 8, "pass_c": 4 * 8}
    if n_dev % 3 == 0 and n_dev // 3 >= 65536 and os.environ.get("FFS_DISABLE_SEGMENTED") != "1" and not args.full_length:
        # block-segmented mode (three length-n/3 blocks per candidate, spectrum products added in the mid
        # pass): mid writes and the last pass reads a third of the candidate slots
        executed = {"pass_a": 4.5 * 8, "mid": (4.5 + 4 / 3) * 8, "pass_c": (4 / 3) * 8}

    def kernel_table(ktimes, steps, n_fft):
        per_kernel = {}
        for k, (ms, n) in ktimes.items():
            if n == 0:
                continue
            pairs_per_launch = P * steps / n
            entry = {"avg_ms": ms / n, "launches": n, "total_ms": ms}
            if k in share:
                entry["algorithmic_bytes_per_launch"] = share[k] * n_ref * pairs_per_launch
                entry["achieved_GBps"] = entry["algorithmic_bytes_per_launch"] / (ms / n * 1e-3) / 1e9
                entry["executed_bytes_per_launch"] = executed[k] * n_fft * pairs_per_launch
                entry["executed_GBps"] = entry["executed_bytes_per_launch"] / (ms / n * 1e-3) / 1e9
            per_kernel[k] = entry
        return per_kernel

    elapsed, ktimes = timed(n_dev, args.steps, args.warmup)

    # correctness of what was timed: recovered offsets/ratios vs the generator's ground truth, and
    # vs the CPU oracle on the sampled pairs below
    pres = pair_out.cpu().numpy().view(_native.PAIR_RESULT_DTYPE).copy()
    cres = cand_out.cpu().numpy().view(_native.CAND_RESULT_DTYPE).reshape(P, n_cand).copy()
    truth_ok = sum(
        int(pres[i]["best_cand"] == sp.true_ratio_index and abs(int(pres[i]["offset"]) - sp.true_offset_samples) <= 30)
        for i, sp in enumerate(specs)
    )
    ambiguous = int(((cres["flags"] & 2) != 0).sum())
    # health of the fp32 transform chain itself (the exact re-evaluation would mask a damaged one as long
    # as the true peak still gets nominated): fp32 value of every winning lag vs its exact score
    fp32_err = float(np.abs(cres["score_f32"].astype(np.float64) - cres["score"]).max())

    solves_per_s = world * P * args.steps / elapsed
    result = {
        "metric": "alignments/sec (2 h@100 Hz, 7 framerate ratios)",
        "value": solves_per_s,
        "unit": "7-ratio solves/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "configs[2]: %d x %.0f s@100 Hz pairs per GPU, MaxScoreAligner over 7 framerate ratios, "
                        "max_offset_samples=6000" % (P, args.duration),
            "n_fft_reference": n_ref,
            "n_fft_device": n_dev,
            "pairs_per_gpu": P,
            "pairs_in_flight": args.pairs_in_flight,
            "parallelism": "pairs sharded by rank, all-gather of 24 B/pair results" if world > 1 else "single GPU",
        },
        "offset_match": {"pairs_matching_ground_truth": truth_ok, "pairs": P, "ambiguous_flags": ambiguous,
                         "max_abs_fp32_error_at_winning_lags": fp32_err},
        "solve_normaliser": {
            "bytes_per_solve": 168 * n_ref,
            "achieved_GBps": solves_per_s * 168 * n_ref / 1e9,
            "frac_of_8TBps_per_gpu": solves_per_s * 168 * n_ref / (HBM_PEAK * world),
            "frac_of_copy_ceiling_per_gpu": solves_per_s * 168 * n_ref / (HBM_COPY_CEILING * world),
        },
    }

    if profile and rank == 0:
        per_kernel = kernel_table(ktimes, args.steps, n_dev)
        dom = max((k for k in per_kernel if k in share), key=lambda k: per_kernel[k]["total_ms"])
        pairs_per_launch = P * args.steps / per_kernel[dom]["launches"]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_per_pair.json")
        if os.path.exists(tpath):  # PMC-measured HBM bytes (profiles/run_pmc.sh), keyed by device length
            tj = json.load(open(tpath)).get(str(n_dev), {})
            if dom in tj:
                traffic = tj[dom] * pairs_per_launch
        result["kernels"] = per_kernel
        result["roofline"] = {
            "kernel": "k_" + dom,
            "bound": "hbm",
            "achieved": per_kernel[dom]["achieved_GBps"],
            "peak": HBM_PEAK / 1e9,
            "unit": "GB/s",
            "frac": per_kernel[dom]["achieved_GBps"] / (HBM_PEAK / 1e9),
            "traffic": traffic,
            "executed_GBps": per_kernel[dom]["executed_GBps"],
            "executed_frac": per_kernel[dom]["executed_GBps"] / (HBM_PEAK / 1e9),
            "note": "achieved = SURVEY 8(d) share of the reference's 168*N(=2^21) bytes per solve / launch time; "
                    "the device moves fewer bytes (packed candidates, reference spectrum kept in registers, "
                    "transform length 3*2^18 instead of 2^21 under the lag window): executed_* uses the bytes this "
                    "kernel really has to move, traffic = PMC-measured HBM bytes per launch",
        }

    if rank == 0 and world == 1 and not args.full_length and not args.skip_full_length_record and n_dev != n_ref:
        # the same batch with the reference's full transform length, for the record
        el2, kt2 = timed(n_ref, max(2, args.steps // 2), 1)
        st2 = max(2, args.steps // 2)
        pres2 = pair_out.cpu().numpy().view(_native.PAIR_RESULT_DTYPE)
        result["full_length"] = {
            "n_fft_device": n_ref,
            "value": P * st2 / el2,
            "ms_per_step": 1e3 * el2 / st2,
            "identical_pair_results": bool(np.array_equal(pres2, pres)),
            "frac_of_8TBps": (P * st2 / el2) * 168 * n_ref / HBM_PEAK,
        }
        if profile:
            result["full_length"]["kernels"] = {k: {kk: v[kk] for kk in ("avg_ms", "achieved_GBps", "executed_GBps") if kk in v}
                                               for k, v in kernel_table(kt2, st2, n_ref).items()}

    if rank == 0 and world == 1 and not args.skip_full_length_record:
        # single-ratio FFTAligner solves/s (BASELINE config 2 as a batch: every pair against
        # the candidate rasterised at its true ratio), with the production lag window and without one
        sdb = db.select_candidates([sp.true_ratio_index for sp in specs])
        single = {}
        for label, mo in (("max_offset_6000", 6000), ("max_offset_none", None)):
            al = batch.BatchAligner(sdb.required_fft_length(mo), 1, max_offset_samples=mo, pairs_in_flight=args.pairs_in_flight)
            al.solve_async(sdb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                c1, p1 = al.solve_async(sdb)
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 3
            c1 = c1.cpu().numpy().view(_native.CAND_RESULT_DTYPE)[:P]
            single[label] = {
                "n_fft_device": int(al.plan.n_fft),
                "solves_per_s": P / el,
                "offsets_within_0.3s_of_truth": int(sum(abs(int(c1[i]["offset"]) - sp.true_offset_samples) <= 30
                                                        for i, sp in enumerate(specs))),
            }
            al.plan.close()
        result["single_ratio"] = single

    if rank == 0 and world == 1 and args.cpu_pairs > 0:
        from oracle import aligners_oracle as orc

        n_s = min(args.cpu_pairs, P)
        inputs = [synth.pair_float_arrays(specs[i]) for i in range(n_s)]
        t1 = time.perf_counter()
        cpu = [orc.max_score_align(r, c, 6000) for r, c in inputs]
        cpu_t = time.perf_counter() - t1
        agree = all(
            int(pres[i]["best_cand"]) == idx and int(pres[i]["offset"]) == off and abs(pres[i]["score"] - sc) <= 1e-5 * abs(sc)
            for i, ((sc, off), idx) in enumerate(cpu)
        )
        result["cpu_baseline"] = {
            "value": n_s / cpu_t,
            "unit": "7-ratio solves/s",
            "cores": 1,
            "kind": "port",
            "sample": "%d of the same pairs (seeds 0..%d), numpy complex128 restatement of aligners.py:50-167 "
                      "(oracle/aligners_oracle.py), single thread, %.1f s" % (n_s, n_s - 1, cpu_t),
            "host_cpus": os.cpu_count(),
        }
        result["offset_match"]["gpu_equals_cpu_oracle_on_sample"] = bool(agree)
        # SURVEY 8(d) baseline (ii): the same restatement on many host cores at once (separate process: it
        # forks workers, which must not happen in a process that has initialised the GPU runtime)
        try:
            import subprocess

            usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            try:  # cgroup v2 CPU quota of this container, if any
                quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
                if quota != "max":
                    usable = min(usable, max(1, int(int(quota) / int(period))))
            except (OSError, ValueError):
                pass
            procs = max(1, min(64, usable // 2 if usable > 2 else usable))
            out = subprocess.run([sys.executable, "-m", "oracle.cpu_parallel_baseline", str(procs), "2", str(args.duration)],
                                 cwd=ROOT, capture_output=True, text=True, timeout=240,
                                 env=dict(os.environ, HIP_VISIBLE_DEVICES="", OMP_NUM_THREADS="1"))
            par = json.loads(out.stdout.strip().splitlines()[-1])
            result["cpu_baseline"]["parallel"] = {
                "value": par["value"], "unit": "7-ratio solves/s", "cores": par["cores"], "usable_cpus": usable,
                "sample": "%d processes (half of the %d CPUs this container may use) x 2 pairs each, same restatement, "
                          "%.1f s per solve per process, %d/%d ratios recovered"
                          % (par["cores"], usable, par["mean_solve_s"], par["recovered"], par["solves"]),
            }
        except Exception as exc:  # a baseline figure must never take the bench line down
            result["cpu_baseline"]["parallel"] = {"error": repr(exc)[:200]}

    if rank == 0 and world == 1 and not args.no_vad:
        result["vad"] = vad_figures(torch, _native)
    if rank == 0 and world == 1 and args.e2e_files > 0:
        result["end_to_end"] = e2e_figures(torch, _native, args.e2e_files)

    if rank == 0:
        print(json.dumps(result))
    if use_dist:
        if rank == 0:
            ok = bool(torch.equal(gathered[rank * P * 24:(rank + 1) * P * 24], pair_out))
            assert ok, "all-gathered records differ from the local ones"
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
