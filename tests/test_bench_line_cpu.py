"""The driver keeps about 10 KB of bench.py's stdout: round 4's line had grown to 24 KB and `BENCH_r04.json.parsed` was null.
bench.py now prints a COMPACT last line (< 4 KB) and writes the full one to bench_full.json.  These checks run the compaction on
the largest full line the repo holds (round 4's, five sub-workloads and the batch sweep folded in), on a padded worst case, and
— in tests/test_mock_device.py::test_bench_dry_run_on_the_mock — on bench.main() itself against the mock device."""
import copy
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _full():
    return json.load(open(os.path.join(ROOT, "profiles", "r4_final_bench_line.json")))


def _check(text, full):
    assert len(text) < 4096, len(text)
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["metric"] == full["metric"]
    rf, cb = line["roofline"], line["cpu_baseline"]
    assert set(rf) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and abs(rf["frac"] - full["roofline"]["frac"]) < 1e-4
    assert abs(rf["achieved"] / rf["peak"] - rf["frac"]) < 1e-3 and rf["avg_launch_ms"] > 0 and rf["launches"] == full["roofline"]["launches"]
    assert set(cb) >= {"value", "unit", "cores", "kind", "sample"} and abs(cb["value"] - full["cpu_baseline"]["value"]) < 1e-3 * cb["value"]
    assert "workload" in line["config"] and line["config"]["rerankK"] == full["config"]["rerankK"]
    return line


def test_compact_line_of_the_round4_full_line():
    import bench
    full = _full()
    assert len(json.dumps(full)) > 20000            # the line that did not survive the driver's capture
    line = _check(bench.compact_line(full), full)
    wl = line["workloads"]
    assert set(wl) >= {"hard_case", "literal_c3", "c2", "c5", "c4_one_shard", "flat_mode"}
    for key, sub in wl.items():
        assert sub["value"] > 0 and sub["unit"], key
        if key != "flat_mode":
            assert abs(sub["value"] - full[key]["value"]) < 1e-3 * sub["value"]
    assert wl["c2"]["roofline_bound"] == "lds" and wl["hard_case"]["rerankK"] == full["hard_case"]["config"]["rerankK"]
    assert wl["c2"]["cpu_value"] > 0 and wl["c5"]["cpu_value"] > 0
    assert line["latency_ms"]["1"] > 0 and line["latency_ms"]["131072"] > 0


def test_compact_line_stays_under_the_limit_when_the_full_line_bloats():
    import bench
    full = _full()
    fat = copy.deepcopy(full)
    fat["config"]["workload"] = fat["config"]["workload"] * 20
    fat["roofline"]["kernel"] = fat["roofline"]["kernel"] * 20
    fat["roofline"]["note"] = "x" * 50000
    fat["cpu_baseline"]["sample"] = fat["cpu_baseline"]["sample"] * 20
    fat["per_rank_qps"] = [1.23456789e6] * 8
    fat["batch_sweep"] = fat["batch_sweep"] * 4
    for k in ("hard_case", "c2"):
        fat[k]["what"] = "y" * 10000
    fat["c5"] = {"error": "rc 1", "stderr_tail": "z" * 600}
    line = _check(bench.compact_line(fat), full)
    assert line["workloads"]["c5"] == {"error": "rc 1"}


def test_emit_writes_the_full_line_and_prints_the_compact_one_last(tmp_path, monkeypatch, capsys):
    import types
    import bench
    full = _full()
    monkeypatch.setenv("JVECTOR_BENCH_FULL", str(tmp_path / "bench_full.json"))
    bench.emit(full, types.SimpleNamespace(sub_line=False))
    cap = capsys.readouterr()
    out = cap.out.strip().splitlines()
    assert len(out) == 1
    line = _check(out[-1], full)
    assert json.load(open(line["full"])) == full and "[full line]" in cap.err
    bench.emit(full, types.SimpleNamespace(sub_line=True))          # a sub-run hands its parent the full line
    assert json.loads(capsys.readouterr().out.strip()) == full
