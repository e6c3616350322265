"""Last / first kernels of one two-stream step of a rocprofv3 kernel trace of bench.py (queue, start, end in us relative to
the step's Adam launch): what the main queue waits for at the end of the backward.
    python scripts/step_tail.py <p_kernel_trace.csv> [step index, default 3] [how many, default 24]"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'at::native' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
for r in rows:
    r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    r['k'] = re.sub(r'\(.*$', '', re.sub(r'^void ', '', r['Kernel_Name'])).replace(' ', '')[:56]
rows.sort(key=lambda r: r['s'])
adam = [i for i, r in enumerate(rows) if r['k'].startswith('adam_kernel')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 24
a = adam[k + 1]
t0 = rows[a]['s']
for r in rows[a - n:a + 8]:
    print('q%-2s %9.1f -> %9.1f (%6.1f)  %s' % (r['Queue_Id'], (r['s'] - t0) / 1e3, (r['e'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r['k']))
