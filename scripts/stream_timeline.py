"""Per-step anatomy of a rocprofv3 kernel trace of `bench.py` (default: two HIP streams): per queue busy time, the
critical-path gaps of the main queue, and how much each kernel family stretches when the two queues overlap.

    python scripts/stream_timeline.py gpurun_out/<dir>/p_kernel_trace.csv [steps]
"""
import collections
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    n = re.sub(r'\(.*$', '', n)
    return n.replace(' ', '')[:60]


rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'at::native' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# one step = from one adam_kernel end to the next
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
if len(adam) < 3:
    raise SystemExit('need at least 3 optimizer steps in the trace')
# (bench.py's last two steps run single-stream for the roofline instrumentation: default to a step of the timed region)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a0, a1 = adam[k], adam[k + 1]
step = rows[a0 + 1:a1 + 1]
t0, t1 = rows[a0]['e'], rows[a1]['e']
print('step wall %.3f ms, %d launches' % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r['Queue_Id']].append(r)
for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r['e'] - r['s'] for r in rs)
    gaps = [(rs[i + 1]['s'] - rs[i]['e']) for i in range(len(rs) - 1)]
    big = sorted(((g, short(rs[i]['Kernel_Name']), short(rs[i + 1]['Kernel_Name'])) for i, g in enumerate(gaps) if g > 20000), reverse=True)
    print('queue %s: %d kernels, busy %.3f ms, span %.3f ms, gaps>20us: %d totalling %.3f ms' % (
        q, len(rs), busy / 1e6, (rs[-1]['e'] - rs[0]['s']) / 1e6, len(big), sum(g for g, _, _ in big) / 1e6))
    for g, a, b in big[:8]:
        print('    gap %.1f us between %s -> %s' % (g / 1e3, a, b))
# overlap: for each kernel, fraction of its duration during which another queue is also busy
iv = [(r['s'], r['e'], r['Queue_Id']) for r in step]
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in step:
    ov = 0
    for s, e, q in iv:
        if q != r['Queue_Id']:
            ov += max(0, min(e, r['e']) - max(s, r['s']))
    f = fam[short(r['Kernel_Name'])]
    f[0] += 1
    f[1] += (r['e'] - r['s']) / 1e3
    f[2] += ov / 1e3
print('%-62s %5s %9s %9s' % ('kernel', 'calls', 'total us', 'overlapped'))
for k, (n, tot, ov) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:22]:
    print('%-62s %5d %9.1f %8.0f%%' % (k, n, tot, 100 * ov / max(tot, 1e-9)))
