"""Print the top rows of a rocprofv3 kernel_stats.csv compactly."""
import csv
import sys

for r in list(csv.DictReader(open(sys.argv[1])))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
    nm = r["Name"].split("(")[0].replace("void gpmi::", "")
    print(f"{nm[:50]:50s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs']) / 1e6:9.2f} ms avg "
          f"{float(r['AverageNs']) / 1e3:9.1f} us {r['Percentage']}%")
