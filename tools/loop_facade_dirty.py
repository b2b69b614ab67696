"""The C++ replan transcription test run N times, each time after this parent has filled most of the device's memory with a
byte pattern and released it: if released memory is not scrubbed, a read of uninitialised memory in either path shows up
as a mismatch.  python tools/loop_facade_dirty.py [runs] [pattern byte, default 255]"""
import os, subprocess, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = "/tmp/frt"
subprocess.check_call(["hipcc", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(root, "pred-occ-planner_amd", "host"),
                       os.path.join(root, "tests", "facade_replan_gpu_test.cpp"), "-o", exe, "-L", os.path.join(root, "pred-occ-planner_amd"),
                       "-lsogm_hip", "-Wl,-rpath," + os.path.join(root, "pred-occ-planner_amd")])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pat = int(sys.argv[2]) if len(sys.argv) > 2 else 255
bad = 0
for i in range(n):
    bufs = []
    try:
        for _ in range(24):
            bufs.append(torch.full((8 << 30,), pat, dtype=torch.uint8, device="cuda"))
    except RuntimeError:
        pass
    torch.cuda.synchronize()
    got = len(bufs)
    del bufs
    torch.cuda.empty_cache()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        bad += 1
        print("run", i, "after", got * 8, "GiB of pattern FAILED:", r.stdout[-600:], r.stderr[-300:], flush=True)
print("failures:", bad, "of", n, "pattern", pat)
