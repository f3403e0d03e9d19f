#!/bin/bash
# One pass over every artefact of a round, from one binary (run on the GPU box through gpurun):
#   GPU tests + smoke; PMC passes at the TIMED launch shape (512 pairs per launch) of the run-boundary kernels (default
#   algorithm) and of the transform kernels (--algorithm fft; default plan + --reference-length); traffic_per_pair.json;
#   rocprofv3 kernel trace of the bench command; rocprofv3 trace + PMC of the secondary kernels (VAD sweep, tokenizer,
#   rasteriser, windowless / reference-length transforms); last the default bench line (reads the fresh traffic file).
#   rocprofv3 kernel trace of the transform path at the timed shape (bench.py --algorithm fft: <TAG>_kernel_stats_fft.csv);
#   TAG=r05 bash profiles/refresh_all.sh
set -u
TAG=${TAG:-r05}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf "$O"; mkdir -p "$O"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" > "$O/${TAG}_gputest.log"; tail -2 "$O/${TAG}_gputest.log"
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
bash profiles/run_pmc.sh "$GRAFT_REPO_ROOT/$O/pmc_runs" > "$O/pmc_runs.log" 2>&1
bash profiles/run_pmc.sh "$GRAFT_REPO_ROOT/$O/pmc_fft" "--algorithm fft" > "$O/pmc_fft.log" 2>&1
bash profiles/run_pmc_tcc.sh "$GRAFT_REPO_ROOT/$O/pmc_ref" "--algorithm fft --reference-length" > "$O/pmc_ref.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python profiles/summarize_pmc.py "$O/pmc_runs" > "$O/${TAG}_pmc_summary_runs.txt"
python profiles/summarize_pmc.py "$O/pmc_fft" > "$O/${TAG}_pmc_summary_n786432.txt"
python profiles/summarize_pmc.py "$O/pmc_ref" > "$O/${TAG}_pmc_summary_n2e21_traffic.txt"
python profiles/make_traffic.py "$O/pmc_fft" 786432 512 > /dev/null
python profiles/make_traffic.py "$O/pmc_runs" 786432 512 > /dev/null
python profiles/make_traffic.py "$O/pmc_ref" 2097152 512 > /dev/null
cp profiles/traffic_per_pair.json "$O/"
bash profiles/run_trace.sh $TAG > "$O/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/trace_$TAG/bench_under_rocprof.json "$O/${TAG}_bench_under_rocprof.json"
find gpurun_out/trace_$TAG -name "*kernel_stats.csv" -exec cp {} "$O/${TAG}_kernel_stats.csv" \;
# the north star's transform kernels at the timed shape (8192 pairs per step, 512 per launch): VERDICT r4 item 7a
bash profiles/run_trace.sh ${TAG}fft "--algorithm fft --steps 6 --warmup 2" > "$O/trace_fft.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/trace_${TAG}fft/bench_under_rocprof.json "$O/${TAG}_bench_under_rocprof_fft.json"
find gpurun_out/trace_${TAG}fft -name "*kernel_stats.csv" -exec cp {} "$O/${TAG}_kernel_stats_fft.csv" \;
rm -rf gpurun_out/trace_${TAG}fft/*/*.db 2>/dev/null
profiles/_bin/lds_atomic_ceiling > "$O/lds_atomic_ceiling.json" 2> /dev/null
for b in lds_issue_rates extract_pattern_ceiling wg_dispatch_rate; do timeout 120 profiles/_bin/$b > "$O/$b.json" 2> /dev/null; done
# secondary kernels: one kernel trace, two PMC passes
S=$GRAFT_REPO_ROOT/$O/secondary
mkdir -p "$S"
( cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$S/trace" -o sec -- python $GRAFT_REPO_ROOT/profiles/secondary_kernels.py 256 > "$S/driver.json" 2> "$S/trace.err"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$S/pmc/f" -o f -- python $GRAFT_REPO_ROOT/profiles/secondary_kernels.py 256 > "$S/pmc_f.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$S/pmc/w" -o w -- python $GRAFT_REPO_ROOT/profiles/secondary_kernels.py 256 > "$S/pmc_w.log" 2>&1 )
python profiles/summarize_secondary.py "$S/trace" "$S/driver.json" "$S/pmc" > "$O/${TAG}_secondary_kernels.json" 2> "$O/secondary.err"
find "$S/trace" -name "*kernel_stats.csv" -exec cp {} "$O/${TAG}_secondary_kernel_stats.csv" \;
rm -rf "$O"/pmc_*/*/*/*.db "$S"/trace/*/*.db "$S"/pmc/*/*/*.db gpurun_out/trace_$TAG/*/*.db 2>/dev/null
rm -rf "$O"/pmc_runs "$O"/pmc_fft "$O"/pmc_ref "$S"/pmc "$S"/trace
python bench.py > "$O/${TAG}_bench.json" 2> "$O/bench.err"; tail -c 400 "$O/${TAG}_bench.json"
du -sh "$O"
