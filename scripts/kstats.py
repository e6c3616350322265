"""Per-kernel ms/step table from a rocprofv3 --kernel-trace --stats run of bench.py:  python scripts/kstats.py <dir>/p_kernel_stats.csv [steps]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 11.0
tot = 0.0
for r in rows:
    ms = float(r['TotalDurationNs']) / 1e6 / steps
    tot += ms
    if ms >= 0.02:
        print('%-110s %5s  %8.1f us  %7.3f ms/step' % (r['Name'][:110], r['Calls'], float(r['AverageNs']) / 1e3, ms))
print('total kernel time %.3f ms/step' % tot)
