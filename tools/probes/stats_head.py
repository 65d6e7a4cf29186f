"""First rows of a rocprofv3 kernel_stats.csv (names contain commas: parsed as CSV)."""
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i >= int(sys.argv[2]) if len(sys.argv) > 2 else 14:
        break
    print(f"{r['Name'][:64]:64s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs']) / 1e6:9.4f} ms  total {float(r['TotalDurationNs']) / 1e6:9.2f} ms")
