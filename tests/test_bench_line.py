"""bench.py's stdout contract: the LAST line is one small JSON object the driver can keep whole (round 4's grew to 27 KB and came
back unparsed).  The full record goes to bench_detail.json / stderr.  The model is the reference's own compact report,
/root/reference/src/retest/reperf.c:804-954 (one short line per measurement)."""
import glob
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full_records():
    out = []
    for p in sorted(glob.glob(os.path.join(ROOT, "profiles", "*bench_default*.json"))):
        try:
            r = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception:
            continue
        if isinstance(r, dict) and "roofline" in r and "metric" in r:
            out.append((os.path.basename(p), r))
    return out


def _synthetic(nsubs=16):
    kern = "fsmhip::walk_generic<fsmhip::CombSelfPol, 1024, true, 3> (mean length < 96 B, decided on the device) | fsmhip::walk_ragged<fsmhip::CombSelfPol, 768, 0>"
    roof = {"bound": "hbm", "achieved": 6341.12, "peak": 8000.0, "unit": "GB/s", "frac": 0.7926, "traffic": 102834567890.5, "traffic_source": "x" * 300,
            "kernel": kern, "kernel_ms_avg": 16.2115, "algorithmic_bytes_per_launch": 102800000000.0,
            "early_retire": {"note": "y" * 400}, "gather_ceiling": {"implied_GBps": 1128.0, "note": "z" * 500}}
    sub = {"workload": "c3_ragged", "value": 3602.11, "unit": "GB/s of line bytes walked", "ms_per_step": 2.8429, "config": {"workload": "w" * 400},
           "roofline": roof, "forms": {f: {"kernel": kern, "note": "n" * 200} for f in ("off64_end", "off32_end", "len_end", "len_bitmap")},
           "cpu_baseline": {"kind": "port", "value": 0.81234, "unit": "GB/s", "cores": 1, "sample": "s" * 300},
           "parity_vs_cpu_sample": "bit-exact", "parity_sample": "p" * 200,
           "full_parity": {"rows": 20000000, "mismatches": 0, "cpu_threads": 16, "seconds": 31.2, "cpu_walk_GBps": 9.1, "checker": "c" * 100}}
    return {"metric": "input GB/s matched (whole node)", "value": 6359.0, "unit": "GB/s", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 16.1036,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "c3: " + "t" * 500, "inputs_per_gpu": 100000000, "input_len": 1024, "dfa_states": 4061, "byte_classes": 40, "table_layout": "combself",
                       "table_bytes": 91000, "sharding": "single GPU", "accepted_inputs": 50000000},
            "roofline": dict(roof, measured_read_stream_GBps=7012.3, frac_of_measured_stream=0.9),
            "cpu_baseline": {"kind": "reference", "value": 0.0041, "unit": "GB/s", "cores": 1, "sample": "s" * 400, "fsm_exec_hoisted_value": 0.31, "vm_v2_value": 0.71,
                             "vm_v2_allcores_value": 10.4, "vm_v2_allcores_cores": 16, "vm_v2_allcores_sample": "q" * 300, "codegen_vmc_value": 0.9, "codegen_vmc_sample": "r" * 300},
            "parity_vs_cpu_sample": "bit-exact", "parity_sample": "p" * 200,
            "full_parity": {"rows": 100000000, "mismatches": 0, "cpu_threads": 16, "seconds": 25.0, "cpu_walk_GBps": 9.0, "checker": "c" * 100},
            "node_front": {"devices": list(range(8)), "uses_rccl": True, "ms_per_step": 17.0, "value_GBps": 48000.0, "front": "f" * 200},
            "sub_results": [dict(sub, workload=f"sub{i}_ragged") for i in range(nsubs)]}


@pytest.mark.parametrize("nsubs", [0, 12, 16, 40])
def test_compact_line_is_small_and_complete(nsubs):
    full = _synthetic(nsubs)
    line = bench.compact_line(full)
    assert "\n" not in line and len(line) < bench.LINE_LIMIT <= 6000
    r = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["value"] == full["value"] and r["ms_per_step"] == full["ms_per_step"] and r["vs_baseline"] is None
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r["roofline"]
    assert r["roofline"]["bound"] in ("hbm", "mfma") and abs(r["roofline"]["frac"] - r["roofline"]["achieved"] / r["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in r["cpu_baseline"]
    assert r["full_parity"]["mismatches"] == 0 and (nsubs > 16 or r["config"]["workload"].startswith("c3"))
    assert len(r.get("sub_results", [])) == nsubs
    for s in r.get("sub_results", []):
        assert s["workload"].startswith("sub") and s["value"] == 3602.11 and s["frac"] == 0.7926 and s["parity"] == "bit-exact"


def test_recorded_full_lines_compact_under_the_limit():
    """the 25-27 KB lines of round 4 (kept under profiles/) now come out small, with the same headline numbers"""
    recs = _full_records()
    assert recs, "no recorded bench lines under profiles/"
    for name, full in recs:
        line = bench.compact_line(full)
        assert len(line) < bench.LINE_LIMIT, (name, len(line))
        r = json.loads(line)
        assert r["value"] == full["value"] and r["roofline"]["frac"] == full["roofline"]["frac"], name
        assert r["cpu_baseline"]["value"] == full["cpu_baseline"]["value"], name
        assert [s["workload"] for s in r.get("sub_results", [])] == [s.get("workload") for s in full.get("sub_results", [])], name


def test_emit_prints_the_compact_line_last_and_writes_the_detail_file(tmp_path, monkeypatch):
    full = _synthetic(12)
    detail = tmp_path / "detail.json"
    monkeypatch.setenv("FSM_BENCH_DETAIL", str(detail))
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        bench.emit(full)
    lines = so.getvalue().strip().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 6000
    r = json.loads(lines[-1])
    assert r["detail"] == str(detail)
    assert json.load(open(detail)) == json.loads(json.dumps(full))           # nothing lost: the whole record is on disk
    assert json.loads(se.getvalue().strip().splitlines()[-1])["sub_results"][0]["forms"]  # ... and on stderr
