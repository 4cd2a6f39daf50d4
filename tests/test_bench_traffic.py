"""bench.py's in-run measurement of `roofline.traffic` (rocprofv3 --pmc around the roofline kernel, after the timed region) must never take the
bench line down: every way it cannot run returns (None, reason) and the line falls back to the committed figure with the reason in
`traffic_source`.  CPU-only: the fallbacks and the rocpd parsing on a synthetic database."""
import importlib.util
import os
import sqlite3
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_refuses_inside_a_profiled_process(monkeypatch):
    b = _bench()
    monkeypatch.setenv("ROCPROF_OUTPUT_PATH", "/tmp/x")
    out, why = b.measure_traffic_pmc()
    assert out is None and "profiled" in why


def test_reports_a_missing_profiler(monkeypatch):
    b = _bench()
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_TOOL")):
            monkeypatch.delenv(k)
    monkeypatch.setattr("shutil.which", lambda name: None)
    real_exists = os.path.exists
    monkeypatch.setattr(os.path, "exists", lambda p: False if p.endswith("rocprofv3") else real_exists(p))
    out, why = b.measure_traffic_pmc()
    assert out is None and "not found" in why


def test_parses_a_rocpd_database(monkeypatch, tmp_path):
    """A fake rocprofv3 that writes the counters_collection table the real one writes: per-dispatch sums over the XCD rows, mean over dispatches."""
    b = _bench()
    for k in list(os.environ):
        if k.startswith(("ROCPROF", "ROCP_TOOL")):
            monkeypatch.delenv(k)
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\nexit 0\n")
    fake.chmod(0o755)
    monkeypatch.setattr("shutil.which", lambda name: str(fake))

    class R:
        returncode = 0
        stderr = b""

    def fake_run(cmd, **kw):
        d = cmd[cmd.index("-d") + 1]
        counter = cmd[cmd.index("--pmc") + 1]
        os.makedirs(os.path.join(d, "host", "1"), exist_ok=True)
        c = sqlite3.connect(os.path.join(d, "host", "1", "pm_results.db"))
        c.execute("create table counters_collection (dispatch_id, kernel_name, counter_name, value, duration)")
        for did in (1, 2):  # two dispatches of the kernel, 8 XCD rows each, plus another kernel that must be ignored
            for x in range(8):
                c.execute("insert into counters_collection values (?,?,?,?,?)", (did, "void gemm_pp4_kernel<1, false, 1>(GemmArgs)", counter, 100.0 * did, 4.0e6))
        c.execute("insert into counters_collection values (?,?,?,?,?)", (3, "at::native::fill", counter, 1e9, 1.0))
        c.commit()
        c.close()
        return R()

    monkeypatch.setattr("subprocess.run", fake_run)
    out, why = b.measure_traffic_pmc()
    assert why is None
    assert out["fetch_kb"] == pytest.approx((800.0 + 1600.0) / 2) and out["write_kb"] == pytest.approx(1200.0)
    assert out["launches"] == 2 and out["dur_us"] == pytest.approx(4000.0)
