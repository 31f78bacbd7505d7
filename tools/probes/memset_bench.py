"""Write-only ceilings on this chip for 404 MB: hipMemsetAsync (torch fill_), a torch copy (read + write), zeros."""
import time, torch
n = 8192 * 156 * 79
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")
def timeit(f, k=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
gb = n * 4 / 1e9
for name, f, traffic in (("fill_(1.5)  [write]", lambda: x.fill_(1.5), 1), ("zero_()     [write]", lambda: x.zero_(), 1),
                         ("copy_       [read+write]", lambda: y.copy_(x), 2), ("mul_(2)     [read+write]", lambda: x.mul_(2.0), 2)):
    us = timeit(f)
    print("%-26s %7.1f us  %7.1f GB/s" % (name, us, traffic * gb / (us * 1e-6)))
