"""Merge the rocprofv3 passes written by scripts/pmc_passes.sh into one per-kernel table (markdown on stdout, JSON with
--json): launches, mean duration, HBM-side traffic (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB counters), achieved GB/s
against the 8 TB/s HBM3E peak, and matrix-pipe utilisation = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE x SIMDs).

    python scripts/pmc_table.py gpurun_out/pmc [--json profiles/rNN_pmc_per_kernel.json]
"""
import collections
import csv
import json
import re
import sys

SIMDS = 256 * 4
HBM_PEAK_GBS = 8000.0


def short(name):
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\(.*$', '', name)
    return name.replace(' ', '')


def read_counters(path):
    """{kernel: {counter: [values per dispatch]}} plus {kernel: [durations ns]} measured under that pass."""
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = short(r['Kernel_Name'])
        vals[k][r['Counter_Name']].append(float(r['Counter_Value']))
        dur[k][r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    return vals, dur


def main():
    root = sys.argv[1]
    fetch, _ = read_counters(root + '/FETCH_SIZE/p_counter_collection.csv')
    write, _ = read_counters(root + '/WRITE_SIZE/p_counter_collection.csv')
    sq, _ = read_counters(root + '/SQ_VALU_MFMA_BUSY_CYCLES/p_counter_collection.csv')
    # un-instrumented durations: the --kernel-trace --stats pass
    dur = collections.defaultdict(list)
    import os
    if os.path.exists(root + '/trace/p_kernel_trace.csv'):
        for r in csv.DictReader(open(root + '/trace/p_kernel_trace.csv')):
            dur[short(r['Kernel_Name'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    else:       # only the --stats summary of that pass was kept: calls x average duration
        for r in csv.DictReader(open(root + '/trace/p_kernel_stats.csv')):
            dur[short(r['Name'])] = [float(r['AverageNs'])] * int(r['Calls'])
    steps_trace = None
    rows = []
    for k, d in dur.items():
        if k.startswith('at::') or 'rocclr' in k:
            continue
        n = len(d)
        mean_us = sum(d) / n / 1e3
        f = fetch.get(k, {}).get('FETCH_SIZE', [])
        w = write.get(k, {}).get('WRITE_SIZE', [])
        fb = 2.0 * 1024 * sum(f) / max(len(f), 1)          # bytes per launch (KiB counter, x2 gfx950 correction)
        wb = 1024.0 * sum(w) / max(len(w), 1)
        busy = sq.get(k, {}).get('SQ_VALU_MFMA_BUSY_CYCLES', [])
        gui = sq.get(k, {}).get('GRBM_GUI_ACTIVE', [])
        util = (sum(busy) / (sum(gui) * SIMDS)) if gui and sum(gui) > 0 else None
        # GRBM_GUI_ACTIVE is summed over the 8 XCDs by rocprofv3: one XCD's count = /8
        util = util * 8 if util is not None else None
        gbs = (fb + wb) / (mean_us * 1e-6) / 1e9
        rows.append(dict(kernel=k, launches=n, mean_us=round(mean_us, 1), total_ms=round(sum(d) / 1e6, 3),
                         fetch_MB=round(fb / 1e6, 1), write_MB=round(wb / 1e6, 1), hbm_GBs=round(gbs, 0),
                         hbm_frac_of_8TBs=round(gbs / HBM_PEAK_GBS, 3),
                         mfma_util=None if util is None else round(util, 3)))
    rows.sort(key=lambda r: -r['total_ms'])
    tot = sum(r['total_ms'] for r in rows)
    print('| kernel | launches | mean us | %% of kernel time | fetch MB | write MB | HBM GB/s | of 8 TB/s | MFMA busy |')
    print('|---|---|---|---|---|---|---|---|---|')
    for r in rows:
        print('| `%s` | %d | %.1f | %.1f | %.1f | %.1f | %.0f | %.0f %% | %s |' % (
            r['kernel'][:70], r['launches'], r['mean_us'], 100 * r['total_ms'] / tot, r['fetch_MB'], r['write_MB'],
            r['hbm_GBs'], 100 * r['hbm_frac_of_8TBs'], '-' if not r['mfma_util'] else '%.0f %%' % (100 * r['mfma_util'])))
    if '--json' in sys.argv:
        json.dump(rows, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)
    if '--traffic' in sys.argv:
        # profiles/pmc_traffic.json: what bench.py's roofline.traffic reports for the dominant kernel, stamped with the
        # fingerprint of the kernel source it was measured on (bench.py reports null when the source has changed since)
        import os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
        import bench
        # every kernel bench.py's timer may name as the dominant one (it reports the symbol with the largest total time of ITS run):
        # the role-split conv instantiations, the weight gradients, the joint backward kernels
        want = [r for r in rows if r['kernel'].startswith(('igemm_pp_kernel', 'wgrad_f16x3_kernel', 'bwd_joint', 'igemm_kernel', 'conv_c1_wgrad'))]
        if not want:
            raise SystemExit('no matrix kernel in the trace')
        kern = {}
        for r in want:
            kern[r['kernel'].replace(' ', '')] = dict(
                bytes_per_launch=int((r['fetch_MB'] + r['write_MB']) * 1e6), fetch_bytes_per_launch=int(r['fetch_MB'] * 1e6),
                write_bytes_per_launch=int(r['write_MB'] * 1e6), launches=r['launches'], mean_us=r['mean_us'],
                mfma_busy_frac=r['mfma_util'])
        out = dict(kernel=want[0]['kernel'].replace(' ', ''), kernel_source_sha=bench.kernel_source_sha(), kernels=kern,
                   method='rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/pmc_passes.sh); KiB counters; '
                          'FETCH_SIZE doubled (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md '
                          'section HBM); WRITE_SIZE as is; single stream, one train step per pass',
                   source=root)
        json.dump(out, open(sys.argv[sys.argv.index('--traffic') + 1], 'w'), indent=1)


if __name__ == '__main__':
    main()
