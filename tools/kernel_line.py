import sys, json
d = json.loads(sys.stdin.readline())
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["ms_per_step"], 3), {k: round(v["ms_per_step"], 3) for k, v in d["kernels"].items()})
