"""Host <-> device copy rates on this box (page-locked and pageable, one direction and both at once): the bound of every
host-memory caller of the path."""
import time
import torch

n = 456_000_000 // 16
dev = torch.device("cuda", 0)
d_in = torch.empty(n, dtype=torch.complex128, device=dev)
d_out = torch.empty(n, dtype=torch.complex128, device=dev)
pin_a = torch.empty(n, dtype=torch.complex128).pin_memory()
pin_b = torch.empty(n, dtype=torch.complex128).pin_memory()
page = torch.empty(n, dtype=torch.complex128)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
gb = n * 16 / 1e9


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


t = timed(lambda: d_in.copy_(pin_a, non_blocking=True))
print(f"H2D page-locked: {gb / t:.1f} GB/s ({t * 1e3:.1f} ms for {gb * 1e3:.0f} MB)")
t = timed(lambda: pin_b.copy_(d_out, non_blocking=True))
print(f"D2H page-locked: {gb / t:.1f} GB/s ({t * 1e3:.1f} ms)")
t = timed(lambda: d_in.copy_(page))
print(f"H2D pageable:    {gb / t:.1f} GB/s ({t * 1e3:.1f} ms)")
t = timed(lambda: page.copy_(d_out))
print(f"D2H pageable:    {gb / t:.1f} GB/s ({t * 1e3:.1f} ms)")


def both():
    with torch.cuda.stream(s1):
        d_in.copy_(pin_a, non_blocking=True)
    with torch.cuda.stream(s2):
        pin_b.copy_(d_out, non_blocking=True)


t = timed(both)
print(f"H2D + D2H at once (two streams, page-locked): {2 * gb / t:.1f} GB/s in all ({t * 1e3:.1f} ms for both)")
