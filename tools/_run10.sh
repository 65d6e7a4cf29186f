REPO=$(pwd); OUT=$REPO/gpurun_out/r02j; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- python3 $REPO/tools/host_mode_rate.py > $OUT/trace.log 2>&1
cd $REPO
python3 - <<'PY'
import csv, glob
k = glob.glob("gpurun_out/r02j/trace/**/*kernel_trace.csv", recursive=True)[0]
m = glob.glob("gpurun_out/r02j/trace/**/*memory_copy_trace.csv", recursive=True)[0]
ev = []
for r in csv.DictReader(open(k)):
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"][:40]))
rows = list(csv.DictReader(open(m)))
print(rows[0].keys())
for r in rows:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
t_end = ev[-1][1]
# last 60 ms of activity
sel = [e for e in ev if e[0] > t_end - 60_000_000 and (e[2] == "C" and (e[1]-e[0]) > 200_000 or e[2] == "K" and (e[1] - e[0]) > 300_000)]
t0 = sel[0][0]
for s, e, kind, name in sel[:80]:
    print(f"{(s - t0) / 1e6:8.2f} {(e - t0) / 1e6:8.2f} {(e - s) / 1e6:6.2f} {kind} {name}")
PY
