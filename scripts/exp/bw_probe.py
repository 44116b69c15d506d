"""Plain streaming bandwidths on this GPU (torch kernels): write-only, read-only, copy."""
import torch, time
n = 256 * 1024 * 1024   # doubles: 2 GiB
a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.5)); print("fill  (write only): %.3f ms  %.2f TB/s" % (ms, n * 8 / ms / 1e9))
ms = t(lambda: a.zero_()); print("zero  (write only): %.3f ms  %.2f TB/s" % (ms, n * 8 / ms / 1e9))
ms = t(lambda: torch.sum(a)); print("sum   (read only):  %.3f ms  %.2f TB/s" % (ms, n * 8 / ms / 1e9))
ms = t(lambda: b.copy_(a)); print("copy  (r+w):        %.3f ms  %.2f TB/s total" % (ms, 2 * n * 8 / ms / 1e9))
ms = t(lambda: torch.add(a, b, out=b)); print("add   (2r+w):       %.3f ms  %.2f TB/s total" % (ms, 3 * n * 8 / ms / 1e9))
