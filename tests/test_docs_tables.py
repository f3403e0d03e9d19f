"""The numbers DESIGN.md section 5 and profiles/README.md quote are GENERATED from the round's committed artefacts
(profiles/make_tables.py); this test fails when the documents and the artefacts disagree, and when the
HIP-event kernel times of the profiled bench runs differ from rocprofv3's by more than 2 % (dominant kernels)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mt():
    spec = importlib.util.spec_from_file_location("make_tables", os.path.join(ROOT, "profiles", "make_tables.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def tables():
    mt = _mt()
    need = [mt.TAG + s for s in ("_kernel_stats.csv", "_kernel_stats_fft.csv", "_bench_under_rocprof.json",
                                 "_bench_under_rocprof_fft.json", "_bench.json")]
    missing = [f for f in need if not os.path.exists(os.path.join(mt.PROF, f))]
    if missing:
        pytest.skip("round artefacts not committed yet: %s" % missing)
    return mt, mt.load()


def test_documents_quote_the_artefacts(tables):
    mt, a = tables
    for path, name, block in ((os.path.join(ROOT, "DESIGN.md"), "design", mt.design_block(a)),
                              (os.path.join(mt.PROF, "README.md"), "readme", mt.readme_block(a))):
        have = mt.current_block(open(path).read(), name)
        assert have is not None, "%s has no generated block '%s'" % (path, name)
        assert have.strip() == block.strip(), "%s: run `python profiles/make_tables.py --write`" % path


def test_hip_event_times_agree_with_rocprofv3(tables):
    mt, a = tables
    rows = {k: (rp, ev) for k, rp, ev in mt.agreement(a)}
    for k in ("k_runs_corr", "k_runs_extract", "k_mid_seg_one", "k_pass_a"):
        rp, ev = rows[k]
        assert abs(rp - ev) <= 0.02 * rp, (k, rp, ev)
    for k, (rp, ev) in rows.items():
        assert abs(rp - ev) <= 0.10 * rp, (k, rp, ev)


def test_design_states_the_test_count_of_the_log(tables):
    mt, a = tables
    n = mt.n_passed(a)
    if n is None:
        pytest.skip("no gputest log")
    assert ("%d passed" % n) in open(os.path.join(ROOT, "DESIGN.md")).read()
