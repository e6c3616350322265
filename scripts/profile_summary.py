"""Assemble profiles/ from the output of scripts/make_profiles.sh:

    python scripts/profile_summary.py gpurun_out/final r01_final
writes profiles/<tag>_bench_train_b16_kernel_stats.csv, <tag>_bench_train_b16_summary.md, <tag>_pmc_per_kernel.json and
refreshes profiles/pmc_traffic.json (HBM-side bytes per launch of bench.py's dominant kernel).
"""
import csv
import io
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
P = os.path.join(ROOT, 'profiles')
bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
infer = json.loads(open(os.path.join(src, 'bench_infer.json')).read().strip().splitlines()[-1])
def _find(root, name):
    for d, _, files in os.walk(root):
        if name in files:
            return os.path.join(d, name)
    return os.path.join(root, name)


stats_csv = _find(os.path.join(src, 'stats'), 'p_kernel_stats.csv')
shutil.copy(stats_csv, os.path.join(P, tag + '_bench_train_b16_kernel_stats.csv'))
rows = list(csv.DictReader(open(stats_csv)))
pmc_json = os.path.join(P, tag + '_pmc_per_kernel.json')
# pmc_traffic.json = what bench.py's roofline.traffic reports, stamped with the kernel-source fingerprint it was measured on
table = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'pmc_table.py'), os.path.join(src, 'pmc'), '--json', pmc_json,
                        '--traffic', os.path.join(P, 'pmc_traffic.json')], capture_output=True, text=True, check=True).stdout
# make_profiles.sh stamps pmc_traffic.json ON THE BOX, before the bench line is taken (so the line carries its own traffic figure):
# that copy is the one to keep
if os.path.exists(os.path.join(src, 'pmc_traffic.json')):
    shutil.copy(os.path.join(src, 'pmc_traffic.json'), os.path.join(P, 'pmc_traffic.json'))
traffic = json.load(open(os.path.join(P, 'pmc_traffic.json')))
traffic['source'] = tag + '_pmc_per_kernel.json'
json.dump(traffic, open(os.path.join(P, 'pmc_traffic.json'), 'w'), indent=1)
for name, dst in (('FETCH_SIZE', 'pmc_fetch_size'), ('WRITE_SIZE', 'pmc_write_size'), ('SQ_VALU_MFMA_BUSY_CYCLES', 'pmc_mfma_busy')):
    f = os.path.join(src, 'pmc', name, 'p_counter_collection.csv')
    if os.path.exists(f):
        shutil.copy(f, os.path.join(P, '%s_%s.csv' % (tag, dst)))
for name in ('bench.json', 'bench_infer.json', 'bench_tta.json', 'bench_w128.json', 'bench_w96.json', 'bench_f32.json'):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(P, '%s_%s' % (tag, name)))
si = _find(os.path.join(src, 'stats_infer'), 'p_kernel_stats.csv')
if os.path.exists(si):
    shutil.copy(si, os.path.join(P, tag + '_bench_infer_b8_kernel_stats.csv'))
st = _find(os.path.join(src, 'stats_tta'), 'p_kernel_stats.csv')
if os.path.exists(st):
    shutil.copy(st, os.path.join(P, tag + '_bench_tta_kernel_stats.csv'))
if os.path.exists(os.path.join(src, 'infer_pmc_per_kernel.json')):      # review item 5: the forward-only command under the three PMC passes
    shutil.copy(os.path.join(src, 'infer_pmc_per_kernel.json'), os.path.join(P, tag.replace('_final', '') + '_infer_pmc_per_kernel.json'))
    shutil.copy(os.path.join(src, 'infer_pmc_table.md'), os.path.join(P, tag.replace('_final', '') + '_infer_pmc_table.md'))

steps = 11   # bench.py --steps 5 --warmup 2 + 4 instrumented steps (2 with, 2 without the side stream)
out = io.StringIO()
out.write('# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline  (%s)\n\n' % tag)
out.write('%d train steps in the trace (2 warm-up + 5 timed + 2 instrumented with the side stream + 2 without), batch 16 of '
          '512x512, 1x MI355X, two HIP streams (kernels of the two streams overlap, so the column sums exceed wall time).\n\n' % steps)
out.write('Un-profiled `python bench.py` on the same box: **%.1f images/s, %.2f ms/step**; dominant kernel `%s`: HIP events '
          '%.1f us/launch = %.1f TF/s algorithmic with the concurrent weight-gradient stream, %.1f TF/s without it.\n\n'
          % (bench['value'], bench['ms_per_step'], bench['roofline']['kernel'], 1e3 * bench['roofline']['avg_launch_ms'],
             bench['roofline']['achieved'], bench['roofline']['achieved_without_concurrent_wgrad_stream']))
out.write('Forward only (`bench.py --mode infer`): %.0f images/s (%s).\n\n' % (infer['value'], infer['config'].get('workload', '')))
out.write('| kernel | calls | avg us | ms/step | % of kernel time |\n|---|---|---|---|---|\n')
for r in rows:
    out.write('| `%s` | %s | %.1f | %.3f | %.1f |\n' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3,
                                                       float(r['TotalDurationNs']) / 1e6 / steps, float(r['Percentage'])))
out.write('\n## PMC passes (single stream, one train step per pass: `scripts/pmc_passes.sh`)\n\n')
out.write('FETCH_SIZE (KiB, doubled: gfx950 tallies 128-B requests at 64 B) and WRITE_SIZE in separate passes; matrix-pipe busy = '
          'sum SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs).  Means over all launches of a kernel in one step.\n\n')
out.write(table)
out.write('\nbench line: `%s`\n' % json.dumps(bench))
open(os.path.join(P, tag + '_bench_train_b16_summary.md'), 'w').write(out.getvalue())
print(out.getvalue()[:3000])
