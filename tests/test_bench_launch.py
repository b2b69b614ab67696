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
