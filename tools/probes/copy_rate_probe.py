"""MEASUREMENT: what a plain device-to-device stream achieves on this GPU (the practical ceiling the HBM-bound passes are held against):
torch copy_ (the runtime's copy kernel), an elementwise a*x+b (read + write) and a pure read (sum) on 2 GB complex128 tensors."""
import time
import torch

dev = torch.device("cuda", 0)
n = 2 * 1024**3 // 16
x = torch.randn(n, dtype=torch.float64, device=dev).to(torch.complex128)
y = torch.empty_like(x)


def rate(fn, bytes_moved, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    return bytes_moved / dt / 1e12, dt * 1e3


nb = x.numel() * 16
print("copy_        : %.2f TB/s (%.3f ms)" % rate(lambda: y.copy_(x), 2 * nb))
print("mul (r + w)  : %.2f TB/s (%.3f ms)" % rate(lambda: torch.mul(x, 2.0, out=y), 2 * nb))
xr = torch.view_as_real(x)
print("sum (read)   : %.2f TB/s (%.3f ms)" % rate(lambda: xr.sum(), nb))
print("fill (write) : %.2f TB/s (%.3f ms)" % rate(lambda: y.zero_(), nb))
