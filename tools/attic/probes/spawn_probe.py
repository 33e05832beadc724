"""Probe: may a process that has initialised the GPU start child processes (subprocess / multiprocessing spawn)?"""
import subprocess, sys, torch
print("cuda:", torch.cuda.is_available(), torch.zeros(1, device="cuda").item())
r = subprocess.run([sys.executable, "-c", "import torch; print('child cuda', torch.zeros(2, device='cuda').sum().item())"], capture_output=True, text=True, timeout=300)
print("child rc", r.returncode, r.stdout.strip(), r.stderr.strip()[-300:])
import torch.multiprocessing as mp
def w(q):
    import torch
    q.put(float(torch.ones(3, device="cuda").sum()))
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    p = ctx.Process(target=w, args=(q,))
    p.start(); print("spawned child says", q.get()); p.join(60); print("exit", p.exitcode)
