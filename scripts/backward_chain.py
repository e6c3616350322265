"""What the backward's two HIP streams actually do (round 3): from a rocprofv3 kernel trace of `bench.py`, for ONE step's
backward, how the matrix kernels -- data gradients (main queue) and weight gradients (side queue) -- and the BatchNorm
passes interleave: matrix-kernel coverage of the span, dgrad/wgrad overlap, and where the BatchNorm passes sit.

    python scripts/backward_chain.py gpurun_out/<dir>/p_kernel_trace.csv [step index, default 3]
"""
import csv
import re
import sys


def short(n):
    n = re.sub(r'^void ', '', n)
    return re.sub(r'\(.*$', '', n).replace(' ', '')


def union(iv):
    iv = sorted(iv)
    out, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            out += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return out + ce - cs


def overlap(a, b):
    tot = 0
    for s, e in a:
        for s2, e2 in b:
            tot += max(0, min(e, e2) - max(s, s2))
    return tot


rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'at::native' not in r['Kernel_Name'] and 'rocclr' not in r['Kernel_Name']]
for r in rows:
    r['s'], r['e'], r['k'] = int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name'])
rows.sort(key=lambda r: r['s'])
# step boundaries: the head kernel (forward's last / backward's first launch) of consecutive steps; a step's backward ends with its
# last adam_kernel (the deferred tail issues two or three Adam launches per step)
heads = [i for i, r in enumerate(rows) if r['k'].startswith('head_fwd_bwd') or r['k'].startswith('head_bwd')]
# bench.py's LAST steps run with one stream / launch by launch (roofline instrumentation): take a step of the timed region
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
h0, h1 = heads[k], heads[k + 1]
last_adam = max(i for i in range(h0, h1) if rows[i]['k'].startswith('adam_kernel'))
prev_adam = max(i for i in range(heads[k - 1], h0) if rows[i]['k'].startswith('adam_kernel'))
fwd = rows[prev_adam + 1:h0]
bwd = rows[h0:last_adam + 1]
step = fwd + bwd
t0, t1 = bwd[0]['s'], bwd[-1]['e']
print('step %.3f ms = forward %.3f + backward (head .. adam) %.3f' % ((bwd[-1]['e'] - rows[prev_adam]['e']) / 1e6, (fwd[-1]['e'] - rows[prev_adam]['e']) / 1e6, (t1 - t0) / 1e6))
is_w = lambda k: k.startswith('wgrad') or k.startswith('conv_c1_wgrad')
is_d = lambda k: (k.startswith('igemm') and True)
W = [(r['s'], r['e']) for r in bwd if is_w(r['k'])]
D = [(r['s'], r['e']) for r in bwd if is_d(r['k'])]
B = [(r['s'], r['e']) for r in bwd if r['k'].startswith('bn_bwd') or r['k'].startswith('maxpool_bwd')]
span = t1 - t0
print('backward span %.3f ms: data gradients %.3f ms (sum), weight gradients %.3f ms (sum), BatchNorm / pool passes %.3f ms (sum)'
      % (span / 1e6, sum(e - s for s, e in D) / 1e6, sum(e - s for s, e in W) / 1e6, sum(e - s for s, e in B) / 1e6))
print('a matrix kernel (dgrad or wgrad) is running during %.1f %% of the span; dgrad and wgrad overlap for %.3f ms (%.1f %% of the dgrad time)'
      % (100.0 * union(W + D) / span, overlap(D, W) / 1e6, 100.0 * overlap(D, W) / max(1, sum(e - s for s, e in D))))
print('BatchNorm / pool passes run beside a weight gradient for %.1f %% of their time, beside a data gradient for %.1f %%'
      % (100.0 * overlap(B, W) / max(1, sum(e - s for s, e in B)), 100.0 * overlap(B, D) / max(1, sum(e - s for s, e in B))))
print('\nfirst 40 matrix / BatchNorm-apply kernels of the backward (us from the head kernel):')
n = 0
for r in bwd:
    if is_w(r['k']) or is_d(r['k']) or r['k'].startswith('bn_bwd_apply_kernel'):
        print('  q%-2s %9.1f -> %9.1f  (%6.1f)  %s' % (r['Queue_Id'], (r['s'] - t0) / 1e3, (r['e'] - t0) / 1e3, (r['e'] - r['s']) / 1e3, r['k'][:64]))
        n += 1
        if n >= 40:
            break

# ---- round 5 (review item 6b): where do the weight-gradient slab reductions (reduce_stage1 / reduce_stage2, side queue) sit
# relative to the MAIN queue -- do they ever hold a data gradient back?  The main queue waits for the side queue only through the
# slot / buffer events recorded behind a weight gradient AND its reductions; a wait shows up as main-queue idle time.
main_q = bwd[0]['Queue_Id']
M = sorted([(r['s'], r['e'], r['k']) for r in bwd if r['Queue_Id'] == main_q])
R = [(r['s'], r['e']) for r in bwd if r['k'].startswith('reduce_stage')]
Wside = [(r['s'], r['e']) for r in bwd if is_w(r['k'])]
gaps = [(M[i][1], M[i + 1][0], M[i + 1][2]) for i in range(len(M) - 1) if M[i + 1][0] - M[i][1] > 0]
idle = sum(e - s for s, e, _ in gaps)
idle_r = overlap([(s, e) for s, e, _ in gaps], R)
idle_w = overlap([(s, e) for s, e, _ in gaps], Wside)
print('\nslab reductions: %d launches, %.3f ms (sum) on the side queue; beside a main-queue kernel for %.1f %% of their time'
      % (len(R), sum(e - s for s, e in R) / 1e6, 100.0 * overlap(R, [(s, e) for s, e, _ in M]) / max(1, sum(e - s for s, e in R))))
print('main queue idle inside the backward: %.3f ms in %d gaps (%.1f %% of the span); a slab reduction is running during %.3f ms of it, '
      'a weight gradient during %.3f ms' % (idle / 1e6, len(gaps), 100.0 * idle / span, idle_r / 1e6, idle_w / 1e6))
big = sorted(gaps, key=lambda g: g[0] - g[1])[:8]
print('largest main-queue gaps (us): ' + ', '.join('%.1f before %s' % ((e - s) / 1e3, k[:28]) for s, e, k in big))
