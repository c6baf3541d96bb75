"""development: why is `secondary.C2_module` (a child of the default bench run) slower than the same command on its own?"""
import json, subprocess, sys, time, os
import torch
CMD = [sys.executable, "bench.py", "--gpus", "1", "--steps", "50", "--warmup", "10", "--no-cpu-baseline", "--no-secondary",
       "--min-seconds", "1.0", "--min-trials", "5", "--config", "C2", "--api", "module"]
def run(tag, **kw):
    r = subprocess.run(CMD, capture_output=True, text=True, **kw)
    d = json.loads(r.stdout.strip().splitlines()[-1]); t = d["trials_ms"]
    print(tag, d["value"], d["ms_per_step"], "trials", len(t), "min", min(t), "first", t[:4], flush=True)
run("no context in parent")
run("no context in parent")
torch.zeros(1, device="cuda"); torch.cuda.synchronize()
run("parent holds a HIP context")
g = torch.cuda.CUDAGraph()
x = torch.zeros(1 << 20, device="cuda")
with torch.cuda.graph(g):
    x.add_(1)
g.replay(); torch.cuda.synchronize()
run("parent holds a context and a graph")
torch.set_num_threads(16)
a = torch.randn(2048, 2048)
t0 = time.time()
while time.time() - t0 < 3: a = (a @ a).tanh()
run("... and has just used a 16-thread CPU pool")
time.sleep(2)
run("... 2 s later")
