"""Host-to-device copy rate of this box for a buffer the size of one packed 256-region batch (107 MB): pinned and pageable, through
torch (hipMemcpyAsync underneath); compare with the library's own h2d time (bench line: h2d_ms)."""
import time
import torch
n = 107 * 1000 * 1000
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, host in (("pinned", torch.empty(n, dtype=torch.uint8).pin_memory()), ("pageable", torch.empty(n, dtype=torch.uint8))):
    host.fill_(1)
    for _ in range(2):
        dev.copy_(host, non_blocking=True); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        dev.copy_(host, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    print("%s: %.2f ms per 107 MB = %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9))
