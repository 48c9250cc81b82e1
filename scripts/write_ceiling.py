"""Measures the practical HBM write ceiling of this MI355X for a 6.27 GB buffer (the bitmap size of configs[2]):
torch's fill / memset / copy kernels. Reference point for k_combine's roofline fraction."""
import torch

n = 1_000_000 * 784
x = torch.empty(n, dtype=torch.int64, device="cuda")
y = torch.empty_like(x)


def timeit(name, fn, bytes_moved):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print(f"{name}: {ms:.3f} ms -> {bytes_moved / ms / 1e6:.0f} GB/s")


timeit("torch.fill_ 6.27GB (write only)", lambda: x.fill_(7), n * 8)
timeit("torch.zero_ 6.27GB (memset)", lambda: x.zero_(), n * 8)
timeit("torch.copy_ 6.27GB (read+write)", lambda: y.copy_(x), 2 * n * 8)
