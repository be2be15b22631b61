"""Host cost of the per-step collation primitives with ONE rank (run under torch.distributed.run)."""
import os, time
import numpy as np, torch, torch.distributed as td
torch.cuda.set_device(0)
td.init_process_group("nccl", device_id=torch.device("cuda", 0))
CAP = 1 << 20
host = torch.zeros(CAP, dtype=torch.uint8).pin_memory(); dev = torch.zeros(CAP, dtype=torch.uint8, device="cuda"); out = torch.zeros(CAP, dtype=torch.uint8, device="cuda")
blob = np.zeros(100000, dtype=np.uint8)
def t(fn, n=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    return round((t1 - t0) / n * 1e6, 1), round((t2 - t0) / n * 1e6, 1)
def stage():
    hv = host.numpy(); hv[:8] = 1; hv[8:8 + blob.size] = blob
def h2d(): dev[:8 + blob.size].copy_(host[:8 + blob.size], non_blocking=True)
w = []
def ag():
    w.append(td.all_gather_into_tensor(out, dev, async_op=True))
def ag_small():
    w.append(td.all_gather_into_tensor(out[:131072], dev[:131072], async_op=True))
def agw():
    td.all_gather_into_tensor(out, dev, async_op=True).wait()
for name, fn in [("stage", stage), ("h2d", h2d), ("all_gather async 1MB", ag), ("all_gather async 128KB", ag_small), ("all_gather+wait", agw)]:
    fn(); print(name, "host us/iter, incl. device drain:", t(fn), flush=True)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    print("on side stream", t(ag), flush=True)
td.destroy_process_group()
