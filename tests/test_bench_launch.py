"""bench.py --gpus N must either run N ranks or refuse: never print an n_gpus = N line from one rank."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_single_gpu_runs_in_process(bench):
    assert bench.launch_plan(1, {}, 1, ["bench.py"]) is None
    assert bench.launch_plan(1, {"RANK": "0", "WORLD_SIZE": "1"}, 8, ["bench.py", "--gpus", "1"]) is None


def test_n_gpus_without_launcher_reexecutes_under_torchrun(bench):
    cmd = bench.launch_plan(8, {}, 8, ["bench.py", "--gpus", "8", "--steps", "20"], port=29555)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    assert cmd[-5:] == ["bench.py", "--gpus", "8", "--steps", "20"]      # the script and its flags, unchanged


def test_under_a_launcher_world_size_must_match(bench):
    assert bench.launch_plan(8, {"RANK": "3", "WORLD_SIZE": "8"}, 8, ["bench.py"]) is None
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.launch_plan(8, {"RANK": "0", "WORLD_SIZE": "2"}, 8, ["bench.py"])
    with pytest.raises(SystemExit, match="local rank 5"):
        bench.launch_plan(8, {"RANK": "5", "WORLD_SIZE": "8", "LOCAL_RANK": "5"}, 4, ["bench.py"])
    # a launcher that shows every rank its own GPU only
    assert bench.launch_plan(8, {"RANK": "5", "WORLD_SIZE": "8", "LOCAL_RANK": "5"}, 1, ["bench.py"]) is None


def test_refuses_more_ranks_than_gpus(bench):
    with pytest.raises(SystemExit, match="only 1 GPU"):
        bench.launch_plan(2, {}, 1, ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit, match="only 0 GPU"):
        bench.launch_plan(1, {}, 0, ["bench.py"])


def _fake_out(flight):
    return {"metric": "m", "value": 14644.88, "value_ok": 13000.1, "unit": "replans/s", "n_gpus": 1, "steps": 20, "warmup": 3,
            "ms_per_step": 8.74, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "w" * 200, "world": "y" * 400, "replans_ok_fraction": 0.9363, "tick_overlap": "z" * 300,
                       "outcomes": {k: 1 for k in "abcdefgh"}},
            "roofline": {"bound": "hbm", "kernel": "k_reset_sectors", "achieved": 2437.6, "peak": 8000.0, "unit": "GB/s", "frac": 0.3047,
                         "traffic": 9.66e8, "traffic_frac": 0.22, "note": "n" * 900,
                         "kernels": [{"kernel": "k_reset_sectors (...)", "x": "p" * 2000},
                                     {"kernel": "k_cull_cylinders + ...", "launch_ms": 0.83}]},
            "cpu_baseline": {"value": 22.8, "unit": "replans/s", "cores": 128, "host_cores": 256, "kind": "port", "sample": "s" * 200,
                             "stages_ms": {"what": "q" * 500}},
            "sustained": {"value": 12270.1, "tick_ms_p50": 9.33, "tick_ms_p99": 23.7, "tick_ms_max": 26.7,
                          "replans_ok_fraction": 0.9034, "ticks": 300, "slowest_tick": {"split_ms": {"ticks": [[0.1] * 5] * 300}}},
            "chain_ms": {"astar_mean": .2, "corridor_mean": 2.1, "corridor_max": 3.2, "qp_mean": 1.74, "qp_max": 4.95,
                         "chain_mean": 4.18, "chain_end": 7.7, "slowest_qp": {"us_per_iteration": 0.887}},
            "variants": {"prestamped_lockstep": {"value": 15570.0, "ms_per_step": 8.22}, "flight": flight},
            "configs": {"cfg4": {"replans_per_s": 10770., "flight": {"replans_per_s": 11200., "flights_failed": 0}},
                        "cfg1": {"ms_per_frame": 3.18},
                        "cfg3_rank_share": {"ms_per_step": 8.37, "projected_cfg3_replans_per_s": 61000., "tick_ms": [8.3] * 300}}}


def test_the_printed_line_ends_with_a_compact_summary(bench, tmp_path, monkeypatch):
    """the driver keeps the last 2000 characters of the line; they must carry every number
    DESIGN section 6 quotes.  The per-tick arrays and the prose go to the detail file."""
    import json

    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    (tmp_path / "profiles").mkdir()
    good = {"value": 18110., "ms_per_step": 7.07, "replans_ok_fraction": .93, "per_agent_tick_ms": {"map": 1.41, "gate_wait": .93},
            "sustained": {"value": 15300., "replans_ok_fraction": 0.9034635, "ms_per_tick_worst_flight": 8.5,
                          "ms_per_tick_best_flight": 8.1}, "flights": 7, "flights_failed": 0}
    line = bench.compact_line(_fake_out(good))
    js = json.dumps(line)
    assert list(line)[-1] == "summary" and len(json.dumps(line["summary"])) <= 1200 and len(js) < 4500
    tail = js[-2000:]
    assert '"summary"' in tail and '"cpu_baseline"' in tail
    s = line["summary"]
    assert s["flight"]["flights_failed"] == 0 and s["flight"]["sust_ok"] == 0.9034635 and s["flight"]["map_gate_ms"] == 2.34
    assert s["lockstep_sustained"]["p99"] == 23.7 and s["cfg3_share"]["ms"] == 8.37 and s["stamp_ms"] == 0.83
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line
    assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(line["roofline"])
    assert {"value", "unit", "cores", "kind", "sample"} <= set(line["cpu_baseline"])
    detail = json.load(open(tmp_path / "profiles" / "bench_last_detail.json"))
    assert len(detail["configs"]["cfg3_rank_share"]["tick_ms"]) == 300     # the arrays live in the file, not in the line
    # a failed flight: no value anywhere in the summary's flight block, the failure count is there
    bad = {"flights": 4, "flights_failed": 1, "error": {"what": "x"}, "sustained": {"ms_per_tick_worst_flight": 52.0,
                                                                                  "ms_per_tick_best_flight": 8.1}}
    s = bench.compact_line(_fake_out(bad))["summary"]["flight"]
    assert s["v"] is None and s["sust_v"] is None and s["flights_failed"] == 1 and s["sust_worst_ms"] == 52.0


def test_a_hung_section_after_the_headline_does_not_cost_the_line(tmp_path):
    """N > 1: the sections behind the headline have never run on more than one real GPU.  If one of them hangs (a rank stuck in a
    collective), rank 0 still prints the line — headline intact, the hung section named, the flight counted as failed — and
    the process leaves by itself with exit code 0."""
    import json
    import subprocess

    (tmp_path / "profiles").mkdir()
    code = f"""
import importlib.util, sys, time
sys.path.insert(0, {os.path.join(ROOT, 'tests')!r})
spec = importlib.util.spec_from_file_location("bench_mod", {os.path.join(ROOT, 'bench.py')!r})
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
from test_bench_launch import _fake_out
m.ROOT = {str(tmp_path)!r}
out = _fake_out(None)
for k in ("sustained", "variants", "configs", "cpu_baseline"):
    out.pop(k)
out["n_gpus"] = 8
dog = m.PostHeadlineWatchdog(out, 0, 0.3)
dog.section = "flight"
time.sleep(30)          # "stuck in a collective"
print("NOT REACHED")
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
    assert r.returncode == 0, r.stderr[-500:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and "NOT REACHED" not in r.stdout
    line = json.loads(lines[0])
    assert line["value"] == 14644.88 and line["n_gpus"] == 8 and "flight" in line["watchdog"]
    assert line["summary"]["flight"]["flights_failed"] == 1 and line["summary"]["flight"]["v"] is None
    # a rank other than 0 leaves silently (later than rank 0); a cancelled watchdog does nothing
    code2 = code.replace("PostHeadlineWatchdog(out, 0, 0.3)", "PostHeadlineWatchdog(out, 1, -14.7)")
    r = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=25)
    assert r.returncode == 0 and r.stdout.strip() == ""
    code3 = code.replace("time.sleep(30)", "dog.cancel(); time.sleep(1)")
    r = subprocess.run([sys.executable, "-c", code3], capture_output=True, text=True, timeout=25)
    assert r.returncode == 0 and r.stdout.strip() == "NOT REACHED"
