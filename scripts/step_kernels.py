"""Kernel launches of ONE train step in launch order from a rocprofv3 kernel trace of bench.py (use DC_STREAMS=1 for
stand-alone durations):  python scripts/step_kernels.py <dir>/p_kernel_trace.csv [step index, default 3] [min us]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n).replace(' ', '')


rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'at::native' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
for r in rows:
    r['s'], r['e'], r['k'] = int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if r['k'].startswith('adam_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
mn = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
step = rows[adam[k] + 1:adam[k + 1] + 1]
t0 = step[0]['s']
tot = 0.0
for r in step:
    d = (r['e'] - r['s']) / 1e3
    tot += d
    if d >= mn:
        print('%9.1f  %8.1f us  q%s  %s' % ((r['s'] - t0) / 1e3, d, r.get('Queue_Id', '?'), r['k'][:90]))
print('step span %.3f ms, kernel time %.3f ms, %d launches' % ((step[-1]['e'] - t0) / 1e6, tot / 1e3, len(step)))
