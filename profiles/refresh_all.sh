#!/bin/bash
# One pass over every round-2 artefact, from one binary (run on the GPU box through gpurun):
#   GPU tests + smoke, PMC passes (default plan + --reference-length), traffic_per_pair.json, rocprofv3 kernel
#   trace of the bench command, and last the default bench line (which reads the fresh traffic file).
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf "$O"; mkdir -p "$O"
python -m pytest tests -m gpu -x -q > "$O/gputest.log" 2>&1; tail -2 "$O/gputest.log"
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
bash profiles/run_pmc.sh "$GRAFT_REPO_ROOT/$O/pmc" > "$O/pmc.log" 2>&1
bash profiles/run_pmc_tcc.sh "$GRAFT_REPO_ROOT/$O/pmc_ref" --reference-length > "$O/pmc_ref.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python profiles/summarize_pmc.py "$O/pmc" > "$O/r02_pmc_summary_n786432.txt"
python profiles/summarize_pmc.py "$O/pmc_ref" > "$O/r02_pmc_summary_n2e21_traffic.txt"
python profiles/make_traffic.py "$O/pmc" 786432 32 > /dev/null
python profiles/make_traffic.py "$O/pmc_ref" 2097152 32 > /dev/null
cp profiles/traffic_per_pair.json "$O/"
bash profiles/run_trace.sh r02 > "$O/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/trace_r02/bench_under_rocprof.json "$O/r02_bench_under_rocprof.json"
find gpurun_out/trace_r02 -name "*kernel_stats.csv" -exec cp {} "$O/r02_kernel_stats.csv" \;
rm -rf "$O"/pmc/*/*/*.db "$O"/pmc_ref/*/*/*.db 2>/dev/null
python bench.py > "$O/r02_bench.json" 2> "$O/bench.err"; tail -c 600 "$O/r02_bench.json"
du -sh "$O"
