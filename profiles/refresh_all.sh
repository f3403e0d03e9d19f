#!/bin/bash
# One pass over every round-3 artefact, from one binary (run on the GPU box through gpurun):
#   GPU tests + smoke, PMC passes at the TIMED launch shape (512 pairs per launch; default plan + --reference-length),
#   traffic_per_pair.json, rocprofv3 kernel trace of the bench command, and last the default bench line (which reads
#   the fresh traffic file).   TAG=r03 bash profiles/refresh_all.sh
set -u
TAG=${TAG:-r03}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh
rm -rf "$O"; mkdir -p "$O"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED|rror" > "$O/${TAG}_gputest.log"; tail -2 "$O/${TAG}_gputest.log"
python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > "$O/smoke.log" 2>&1; tail -1 "$O/smoke.log"
bash profiles/run_pmc.sh "$GRAFT_REPO_ROOT/$O/pmc" > "$O/pmc.log" 2>&1
bash profiles/run_pmc_tcc.sh "$GRAFT_REPO_ROOT/$O/pmc_ref" --reference-length > "$O/pmc_ref.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python profiles/summarize_pmc.py "$O/pmc" > "$O/${TAG}_pmc_summary_n786432.txt"
python profiles/summarize_pmc.py "$O/pmc_ref" > "$O/${TAG}_pmc_summary_n2e21_traffic.txt"
python profiles/make_traffic.py "$O/pmc" 786432 512 > /dev/null
python profiles/make_traffic.py "$O/pmc_ref" 2097152 512 > /dev/null
cp profiles/traffic_per_pair.json "$O/"
bash profiles/run_trace.sh $TAG > "$O/trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"
cp gpurun_out/trace_$TAG/bench_under_rocprof.json "$O/${TAG}_bench_under_rocprof.json"
find gpurun_out/trace_$TAG -name "*kernel_stats.csv" -exec cp {} "$O/${TAG}_kernel_stats.csv" \;
rm -rf "$O"/pmc/*/*/*.db "$O"/pmc_ref/*/*/*.db 2>/dev/null
rm -rf "$O"/pmc "$O"/pmc_ref gpurun_out/trace_$TAG/*/*.db 2>/dev/null
python bench.py > "$O/${TAG}_bench.json" 2> "$O/bench.err"; tail -c 600 "$O/${TAG}_bench.json"
du -sh "$O"
