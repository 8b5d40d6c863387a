"""Print the top rows of a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats file."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:top]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), r["AverageNs"].rjust(12), r["Percentage"].rjust(7))
