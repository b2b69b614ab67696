"""The C++ replan transcription test (tests/facade_replan_gpu_test.cpp) run N times as the child of a process that holds a
GPU context with streams of its own (what the pytest process is to it): counts the runs whose per-object path does not
reproduce the fused replan bit for bit.  python tools/loop_facade_parent.py [runs]"""
import os, subprocess, sys
import torch
x = torch.zeros(1 << 20, device="cuda")
streams = [torch.cuda.Stream() for _ in range(24)]
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = "/tmp/frt"
subprocess.check_call(["hipcc", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(root, "pred-occ-planner_amd", "host"),
                       os.path.join(root, "tests", "facade_replan_gpu_test.cpp"), "-o", exe, "-L", os.path.join(root, "pred-occ-planner_amd"),
                       "-lsogm_hip", "-Wl,-rpath," + os.path.join(root, "pred-occ-planner_amd")])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for i in range(n):
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        bad += 1
        print("run", i, "FAILED:", r.stdout[-700:], flush=True)
print("failures:", bad, "of", n)
