"""The bench line the driver parses: the LAST stdout line of bench.py must be one JSON object below 4 KB (the driver keeps
an 8 KB tail of stdout; round 4's 21.6 KB single line left `BENCH_r04.json.parsed` null).  CPU-only: the compact record
is built from a full record -- here the committed round-4 one -- without touching a GPU."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _full_record():
    lines = [l for l in open(os.path.join(ROOT, "profiles", "r04_bench.json")) if l.startswith("{")]
    return json.loads(lines[-1])


def test_compact_line_is_short_and_carries_the_contract():
    full = _full_record()
    line = bench.compact_record(full, "bench_detail.json")
    assert "\n" not in line and len(line) < 4096
    rec = json.loads(line)
    for k in CONTRACT:
        assert k in rec, k
    assert rec["value"] == float("%.6g" % full["value"])
    assert set(("workload", "algorithm", "path", "pairs_per_gpu", "pairs_in_flight")) <= set(rec["config"])
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(rec["roofline"])
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(rec["cpu_baseline"])
    assert rec["offset_match"]["pairs_matching_reference_golden"] == "1024/1024"
    for k in ("fft_path_value", "reference_length_value", "windowless_value"):
        assert isinstance(rec[k], float)


def test_compact_line_stays_short_when_the_record_grows():
    full = _full_record()
    full["config"]["workload"] = "w" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    full["kernels"] = {"k%d" % i: {"us_per_pair": 0.123456789 * i} for i in range(400)}
    full["boundary_density"]["sweep"] = full["boundary_density"]["sweep"] * 40
    line = bench.compact_record(full, "bench_detail.json")
    assert len(line) < 4096
    rec = json.loads(line)
    for k in CONTRACT:
        assert k in rec, k


def test_emit_ends_with_the_compact_line(tmp_path, monkeypatch):
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    buf = io.StringIO()
    with redirect_stdout(buf):
        bench.emit(_full_record())
    out = buf.getvalue().rstrip("\n").split("\n")
    assert out[0].startswith("# detail: {")
    last = out[-1]
    assert len(last) < 4096 and len("\n".join(out)[-8192:].split("\n")[-1]) == len(last)
    assert json.loads(last)["detail_file"] == "bench_detail.json"
    assert json.loads(open(tmp_path / "bench_detail.json").read())["value"] == _full_record()["value"]
