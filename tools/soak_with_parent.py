"""Soak under a second GPU process: this parent initialises HIP (a context, 24 streams with a kernel each — what a pytest
process that ran GPU tests holds) and stays alive while tools/soak_queues.py flies in a child process."""
import os, subprocess, sys
import torch
x = torch.zeros(1 << 20, device="cuda")
streams = [torch.cuda.Stream() for _ in range(24)]
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_queues.py")] + sys.argv[1:], capture_output=True, text=True)
print(r.stdout[-1500:], r.stderr[-300:] if r.returncode else "")
