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


def test_reference_api_header_compiles(tmp_path):
    """host/sogm_reference_api.hpp (per-object shims with the reference's signatures) compiles against the ABI and
    links; the transcription of FakeBaselinePlanner::replan built on it runs in the GPU test below."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "facade_replan_gpu_test")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-ffp-contract=off", "-I",
                           os.path.join(ROOT, "pred-occ-planner_amd", "host"),
                           os.path.join(ROOT, "tests", "facade_replan_gpu_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "pred-occ-planner_amd"), "-lsogm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "pred-occ-planner_amd")])
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_replan_transcription_on_gpu(tmp_path):
    """FakeBaselinePlanner::replan (baseline_fake.cpp:266-472) and BaselinePlanner::replan (baseline.cpp:252-450)
    transcribed statement by statement against the per-object shims (search / getPathWithVel / getObstaclePoints / firi::firi / ShrinkCorridor / the LP checks /
    BezierOpt / isSafeAfterOpt), three agents: every verdict and every control point equals the fused sogm_replan."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "facade_replan_gpu_test")
    subprocess.check_call([hipcc, "-std=c++17", "-O1", "-ffp-contract=off", "-I",
                           os.path.join(ROOT, "pred-occ-planner_amd", "host"),
                           os.path.join(ROOT, "tests", "facade_replan_gpu_test.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "pred-occ-planner_amd"), "-lsogm_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "pred-occ-planner_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)
    assert "facade replan transcription ok" in out.stdout
    print(out.stdout)
