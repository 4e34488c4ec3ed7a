"""Turn gpurun_out/r2prof (profiles/tools/collect_r2.sh) into the committed
profiles/r2_* files.  Run in the authoring container after the gpurun call."""
import collections
import csv
import json
import os
import shutil

O = 'gpurun_out/r2prof'
out = []
P = out.append


def stats(tag, title, rows_shown=4):
  P('## rocprofv3 --kernel-trace --stats --output-format csv -- ' + title)
  P('%-88s %6s %14s %13s %8s %12s %12s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'pct',
                                            'min_ns', 'max_ns'))
  rows = list(csv.DictReader(open(O + '/prof_%s/bench_kernel_stats.csv' % tag)))
  for r in rows[:rows_shown]:
    P('%-88s %6s %14s %13.0f %8s %12s %12s' % (r['Name'][:88], r['Calls'], r['TotalDurationNs'],
                                               float(r['AverageNs']), r['Percentage'],
                                               r['MinNs'], r['MaxNs']))
  return rows[0]


def pmc(tag, pattern):
  acc = collections.defaultdict(list)
  last = None
  for r in csv.DictReader(open(O + '/%s/pmc_counter_collection.csv' % tag)):
    if pattern in r['Kernel_Name']:
      acc[r['Counter_Name']].append((float(r['Counter_Value']),
                                     int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
      last = r
  return acc, last


P('# Round 2 -- measurement run (profiles/tools/collect_r2.sh, one gpurun call, 1 x MI355X)')
P('# raw outputs: gpurun_out/r2prof (scratch); this file: profiles/tools/summarize_r2.py\n')
P(open(O + '/smoke.txt').read().strip() + '\n')

for tag, title in (('B4096', 'python bench.py --secondary-batch 0 --warmup 1000 --cpu-seconds 0   '
                             '(headline: Burgers N=64, B=4096, 1000 midpoint steps, persistent)'),
                   ('B1024', 'python bench.py --batch 1024 --secondary-batch 0 --warmup 1000 --cpu-seconds 0   '
                             '(BASELINE configs[1])')):
  top = stats(tag, title)
  d = json.load(open(O + '/bench_%s.json' % tag))
  r = d['roofline']
  P('(every launch of the kernel is 1000 steps long: pre-heat launches, the --warmup launch and the timed one(s);')
  P(' bench.py HIP-event time per timed launch in the un-profiled run of the same command: %.3f ms -> %.2f TFLOP/s fp32 '
    '= %.1f %% of 157.3;' % (r['kernel_ms_per_launch'], r['achieved'], 100 * r['frac']))
  P(' rocprofv3 average / minimum over %s launches: %.3f / %.3f ms; sclk %s)\n' % (
      top['Calls'], float(top['AverageNs']) / 1e6, float(top['MinNs']) / 1e6,
      json.dumps(d.get('clocks'))))
  shutil.copy(O + '/bench_%s.json' % tag, 'profiles/r2_final_bench_%s.json' % tag)

top = stats('persub', 'python bench.py --launch-mode per_substep --secondary-batch 0 --steps 200 --warmup 200 '
                      '--preheat-ms 50 --cpu-seconds 0   (one fused launch per RK substep, B=4096)')
d = json.load(open(O + '/bench_persub.json'))
P('(substep_multi_kernel; bench.py (1000 steps, un-profiled): %.4f ms per launch, %.2f TFLOP/s = %.1f %%; the kernel\'s own '
  'duration under rocprofv3 is %.4f ms: the overhead against the persistent integrator (%.4f ms per stage) is inside the '
  'launch -- cold L2 after the kernel boundary, weight / table fetch before the first MFMA, no evaluation to overlap the '
  'first and last phases with -- not between launches)\n' % (
      d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved'], 100 * d['roofline']['frac'],
      float(top['AverageNs']) / 1e6,
      json.load(open(O + '/bench_B4096.json'))['roofline']['kernel_ms_per_launch'] / 2000))
shutil.copy(O + '/bench_persub.json', 'profiles/r2_final_bench_persub_B4096.json')

top = stats('stream', 'python bench.py --equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 '
                      '--secondary-batch 0 --steps 100 --warmup 100 --preheat-ms 50 --cpu-seconds 0')
d = json.load(open(O + '/bench_stream.json'))
P('(stream::fixed_substep_kernel, the HBM-shaped kernel; bench.py: %.4f ms per launch, %.0f GB/s algorithmic = %.1f %% of 8 TB/s)\n' % (
    d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved'], 100 * d['roofline']['frac']))
shutil.copy(O + '/bench_stream.json', 'profiles/r2_final_bench_fixed_kdv_persub.json')

for batch in (4096, 1024):
  acc, last = pmc('pmc_sq_B%d' % batch, 'integrate_kernel')
  evals = batch * 2000
  dur = min(v[1] for v in acc['SQ_WAVE_CYCLES'])
  P('## PMC (rocprofv3 --kernel-trace --pmc SQ_*): bench.py --batch %d --secondary-batch 0 --steps 1000 --warmup 0 '
    '--preheat-ms 0 --min-timed-ms 0' % batch)
  P('## %d wavefronts x 2000 RHS evaluations; kernel %.3f ms under the profiler (fastest of %d dispatches); grid %s, '
    'workgroup %s, LDS %s B, scratch %s, VGPRs %s' % (
        batch, dur / 1e6, len(acc['SQ_WAVE_CYCLES']), last['Grid_Size'], last['Workgroup_Size'],
        last['LDS_Block_Size'], last['Scratch_Size'], last['VGPR_Count']))
  per = {}
  for name, values in sorted(acc.items()):
    value = min(v[0] for v in values)
    quad = name not in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_BUSY_CYCLES')
    unit = ('cycles (quad-cycles x4)' if quad else 'instructions' if 'INSTS' in name else 'cycles')
    per[name] = value * (4 if quad else 1) / evals
    P('  %-26s total %.4g   per wave-evaluation %8.0f %s' % (name, value, per[name], unit))
  waves_per_simd = 2 if batch >= 2048 else 1
  P('  %d wavefront(s) per SIMD: %d x %.0f MFMA cycles / %.0f wave cycles = %.1f %% by these counters (SQ_WAVE_CYCLES is a sampled '
    'quad-cycle count; the wall-clock figure below is the one quoted); MFMA share of issued VALU instructions %d of %.0f'
    % (waves_per_simd, waves_per_simd, per['SQ_VALU_MFMA_BUSY_CYCLES'], per['SQ_WAVE_CYCLES'],
       100 * waves_per_simd * per['SQ_VALU_MFMA_BUSY_CYCLES'] / per['SQ_WAVE_CYCLES'], 651,
       per['SQ_INSTS_VALU']))
  P('  useful share of the MFMA cycles: 434 304 FMA per wave-evaluation / (14 616 cycles x 32 FMA per cycle) = 92.9 %\n')

P('## shader clock under the kernel (independent readings)')
clock = {}
for batch in (4096, 1024):
  acc, last = pmc('pmc_clk_B%d' % batch, 'integrate_kernel')
  rows = acc['GRBM_GUI_ACTIVE']
  best = min(rows, key=lambda v: v[1])
  clock[batch] = best[0] / 8 / best[1] * 1e3     # the counter is summed over the 8 XCDs
  P('  B=%d: GRBM_GUI_ACTIVE %.4g cycles (sum over 8 XCDs) over %.3f ms (fastest of %d launches) -> %.0f MHz' % (
      batch, best[0], best[1] / 1e6, len(rows), clock[batch]))
  evals_per_simd = (batch // 1024) * 2000
  cycles = best[1] * 1e-9 * clock[batch] * 1e6 / evals_per_simd
  P('        -> %.0f shader cycles per wave-evaluation slot on every SIMD; matrix pipe busy %.1f %% of wall time '
    '(14 616 MFMA cycles per evaluation); x 92.9 %% useful x %.0f / 2400 MHz = %.1f %% of the 157.3 TFLOP/s peak' % (
        cycles, 100 * 14616 / cycles, clock[batch], 100 * 14616 / cycles * 0.9286 * clock[batch] / 2400))
for batch in (4096, 1024):
  text = open(O + '/phases_B%d.txt' % batch).read()
  for line in text.splitlines():
    if 'launch per wave' in line:
      P('  B=%d traced kernel: %s' % (batch, line.strip()))
  d = json.load(open(O + '/bench_B%d.json' % batch))
  P('  B=%d sysfs during the timed region: %s' % (batch, json.dumps(d.get('clocks'))))
P('')

P('## HBM traffic (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes; KB per dispatch)')
traffic = {}
for tag in ('B4096', 'B1024', 'persub', 'stream'):
  pattern = {'persub': 'substep_multi_kernel', 'stream': 'fixed_substep'}.get(tag, 'integrate_kernel')
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc, _ = pmc('pmc_%s_%s' % (counter, tag), pattern)
    v = [x[0] for x in acc[counter]]
    traffic[(tag, counter)] = sum(v) / len(v)
    P('  %-8s %-10s dispatches %4d  mean %10.1f KB  min %10.1f  max %10.1f' % (
        tag, counter, len(v), sum(v) / len(v), min(v), max(v)))
P('  B4096 / B1024 (persistent integrator): y0 + forcing rows (80 B x 20 modes per sample) + weights read, one snapshot')
P('    written: algorithmic; 4-byte-per-lane loads, FETCH_SIZE = bytes.  persub: per substep, state in + base + out.')
P('  stream (fixed stencils, B=262144 x N=64 = 64 MiB per array): FETCH_SIZE alternates ~32768 / ~65536 KB between the two')
P('    midpoint stages = HALF of what the float4 loads fetch (gfx950 reports 1/2 for 16-byte-per-lane streams,')
P('    MI355X_MICROARCH.md): corrected traffic = algorithmic bytes.\n')

for batch in (1024, 4096):
  P('## s_memtime phase trace (profiles/tools/trace_phases.py %d; dedicated traced instantiation)' % batch)
  P(open(O + '/phases_B%d.txt' % batch).read().rstrip())
P('  (pure matrix-pipe time: input + hidden 10 752, output layer 3 864 cycles)')

open('profiles/r2_final_rocprof_summary.txt', 'w').write('\n'.join(out) + '\n')

entries = []
def entry(tag, match, correction, note):
  f, w = traffic[(tag, 'FETCH_SIZE')], traffic[(tag, 'WRITE_SIZE')]
  entries.append(dict(match=match, fetch_size_kb=f, write_size_kb=w, fetch_correction=correction,
                      traffic_bytes_per_launch=int(round(1024 * (correction * f + w))), note=note))
base = dict(equation='ConservativeBurgersEquation', num_points=64, fixed=False)
entry('B4096', dict(base, batch_per_gpu=4096, launch_mode='persistent'), 1.0,
      'persistent integrator, any --steps (one snapshot): y0 + forcing rows + weights in, final state out')
entry('B1024', dict(base, batch_per_gpu=1024, launch_mode='persistent'), 1.0,
      'persistent integrator, any --steps (one snapshot)')
entry('persub', dict(base, batch_per_gpu=4096, launch_mode='per_substep'), 1.0,
      'substep_multi_kernel, mean over the two midpoint stages')
entry('stream', dict(equation='ConservativeKdVEquation', num_points=64, batch_per_gpu=262144,
                     launch_mode='per_substep', fixed=True), 2.0,
      'stream_fixed kernel, mean over the two midpoint stages; 16-byte-per-lane loads: FETCH_SIZE '
      'reports half the bytes on gfx950, x2 applied')
json.dump(dict(source='gpurun_out/r2prof/pmc_{FETCH,WRITE}_SIZE_* (round 2, profiles/tools/collect_r2.sh), '
                      'summarized in profiles/r2_final_rocprof_summary.txt', entries=entries),
          open('profiles/r2_hbm_traffic.json', 'w'), indent=1)
print('\n'.join(out))
