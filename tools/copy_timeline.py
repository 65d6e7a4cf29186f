"""Summarise a rocprofv3 --memory-copy-trace --kernel-trace run of tools/host_mode_rate.py: for the LAST transform of the run, the busy
intervals of uploads, downloads (memory copies and the runtime's blit kernels) and compute kernels, and how much they overlap."""
import csv, glob, sys

root = sys.argv[1]
def load(pattern):
    rows = []
    for f in glob.glob(root + "/**/" + pattern, recursive=True):
        rows += list(csv.DictReader(open(f)))
    return rows
copies = load("*memory_copy_trace.csv")
kernels = load("*kernel_trace.csv")
ev = []
for r in copies:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Kind", "?"))))
for r in kernels:
    name = r["Kernel_Name"]
    kind = "blit" if "copyBuffer" in name or "rocclr" in name else "kernel"
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind))
ev.sort()
t_end = ev[-1][1]
win = [e for e in ev if e[0] > t_end - 16_000_000]  # the last 16 ms
def busy(kind_prefix):
    iv = sorted((a, b) for a, b, k in win if k.startswith(kind_prefix))
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None: tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None: tot += cur_b - cur_a
    return tot / 1e6, len(iv)
kinds = sorted({k for _, _, k in win})
print("window: last 16 ms of the trace; busy time per kind (ms, count):")
for k in kinds:
    print("  ", k, busy(k))
sizes = {}
for r in copies:
    sizes.setdefault(r.get("Direction", "?"), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in sizes.items():
    print("copy durations", k, "n =", len(v), "max ms", max(v), "sum ms", sum(v))
