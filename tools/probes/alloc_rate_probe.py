"""What a large device allocation costs on this box: hipMalloc (through torch's allocator, cache emptied), first touch, free."""
import time, torch
torch.cuda.init()
torch.empty(1, device="cuda")
def one(gb, keep=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    x = torch.empty(int(gb * 2**30), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    x.zero_()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{gb:5.1f} GB{' (held)' if keep is not None else ''}: malloc {1e3 * (t1 - t0):8.1f} ms, first touch {1e3 * (t2 - t1):8.1f} ms", flush=True)
    if keep is not None:
        keep.append(x)
    else:
        del x
        torch.cuda.empty_cache()
for gb in (1, 8, 24, 28, 31.9, 32.1, 36, 40, 48, 64, 48, 96):
    one(gb)
print("held allocations of 24 GB, one after the other:")
held = []
for _ in range(6):
    one(24, held)
