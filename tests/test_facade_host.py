"""Wire-format helpers of the C++ facade (row f4: BezierTraj.msg <-> SogmTrajRecord, map/future_risk layout):
compiled with hipcc (host side only) and run on the CPU; also proves the header compiles against the ABI."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_facade_wire_formats(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "facade_host_test")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "pred-occ-planner_amd", "host"),
                           os.path.join(ROOT, "tests", "facade_host_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "pred-occ-planner_amd"), "-lsogm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "pred-occ-planner_amd")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "facade host ok" in out.stdout


@pytest.mark.gpu
def test_facade_end_to_end_on_gpu(tmp_path):
    """The reference's call sequence (map update, getClearOcccupancy, getObstaclePoints, search, corridors, optimise,
    replan) written in C++ against the facade classes, compiled with hipcc and run on the GPU."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "facade_gpu_test")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-I", os.path.join(ROOT, "pred-occ-planner_amd", "host"),
                           os.path.join(ROOT, "tests", "facade_gpu_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "pred-occ-planner_amd"), "-lsogm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "pred-occ-planner_amd")])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "facade gpu ok" in out.stdout
