"""Every launch of ONE train step of a rocprofv3 kernel trace of `bench.py`, in start order, per queue: start (us from the step's first
kernel), duration, gap to the previous kernel of the SAME queue.  The small-window steps (128^2 x 20) are bound by this chain.

    python scripts/queue_timeline.py gpurun_out/<dir>/p_kernel_trace.csv [step index, default 8]
"""
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
heads = [i for i, r in enumerate(rows) if r['k'].startswith('head_fwd_bwd') or r['k'].startswith('head_bwd')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
h0 = heads[k]
prev_adam = max(i for i in range(heads[k - 1], h0) if rows[i]['k'].startswith('adam_kernel'))
last_adam = max(i for i in range(h0, heads[k + 1]) if rows[i]['k'].startswith('adam_kernel'))
step = rows[prev_adam + 1:last_adam + 1]
t0 = rows[prev_adam]['e']
main_q = rows[h0]['Queue_Id']
last = {}
tot = {}
print('step %.1f us, %d launches; main queue = q%s' % ((step[-1]['e'] - t0) / 1e3, len(step), main_q))
for r in step:
    q = r['Queue_Id']
    gap = (r['s'] - last[q]) / 1e3 if q in last else (r['s'] - t0) / 1e3
    last[q] = r['e']
    d = tot.setdefault(q, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += (r['e'] - r['s']) / 1e3
    d[2] += max(gap, 0.0)
    print('%s %9.1f  dur %7.1f  gap %7.1f  %s' % ('M' if q == main_q else '  s', (r['s'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, gap, r['k'][:70]))
for q, d in tot.items():
    print('queue %s%s: %d launches, busy %.1f us, gaps %.1f us' % (q, ' (main)' if q == main_q else '', d[0], d[1], d[2]))

if len(sys.argv) > 3 and sys.argv[3] == 'all':
    # every step of the trace: span, main-queue idle time, largest main-queue gap
    print('\nstep  span us  main idle us  largest main gap')
    for kk in range(1, len(heads) - 1):
        try:
            pa = max(i for i in range(heads[kk - 1], heads[kk]) if rows[i]['k'].startswith('adam_kernel'))
            la = max(i for i in range(heads[kk], heads[kk + 1]) if rows[i]['k'].startswith('adam_kernel'))
        except ValueError:
            continue
        st = [r for r in rows[pa + 1:la + 1] if r['Queue_Id'] == main_q]
        gaps = [(st[i + 1]['s'] - st[i]['e']) / 1e3 for i in range(len(st) - 1)]
        print('%4d  %7.1f  %7.1f  %7.1f' % (kk, (rows[la]['e'] - rows[pa]['e']) / 1e3, sum(g for g in gaps if g > 0), max(gaps)))
