"""Turn gpurun_out/r6prof (profiles/tools/collect_r6.sh) into the committed
profiles/r6_* files.  Run in the authoring container after the gpurun call."""
import collections
import csv
import glob
import json
import os
import shutil

O = 'gpurun_out/r6prof'
out = []
P = out.append


def find(tag, suffix):
  hits = glob.glob(os.path.join(O, tag, '**', '*' + suffix), recursive=True)
  return hits[0] if hits else None


def stats(tag, title, rows_shown=6, keep=None):
  path = find(tag, 'kernel_stats.csv')
  P('## rocprofv3 --kernel-trace --stats --output-format csv -- ' + title)
  if path is None:
    P('   (no kernel_stats.csv: see %s.log)' % tag)
    return []
  P('%-96s %7s %14s %12s %7s %11s %11s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'pct',
                                           'min_ns', 'max_ns'))
  rows = list(csv.DictReader(open(path)))
  shown = 0
  for r in rows:
    if keep is not None and not any(k in r['Name'] for k in keep):
      continue
    P('%-96s %7s %14s %12.0f %7s %11s %11s' % (r['Name'][:96], r['Calls'], r['TotalDurationNs'],
                                               float(r['AverageNs']), r['Percentage'],
                                               r['MinNs'], r['MaxNs']))
    shown += 1
    if shown >= rows_shown:
      break
  return rows


def pmc(tag, pattern):
  path = find(tag, 'counter_collection.csv')
  acc = collections.defaultdict(list)
  last = None
  if path is None:
    return acc, last
  for r in csv.DictReader(open(path)):
    if pattern in r['Kernel_Name']:
      acc[r['Counter_Name']].append((float(r['Counter_Value']),
                                     int(r['End_Timestamp']) - int(r['Start_Timestamp']),
                                     r.get('Grid_Size')))
      last = r
  return acc, last


P('# Round 6 -- measurement run (profiles/tools/collect_r6.sh, one gpurun call, 1 x MI355X)')
P('# raw outputs: gpurun_out/r6prof (scratch); this file: profiles/tools/summarize_r6.py\n')
P(open(O + '/smoke.txt').read().strip() + '\n')

line = json.load(open(O + '/bench_default.json'))
shutil.copy(O + '/bench_default.json', 'profiles/r6_bench_default.json')
P('## python bench.py   (the driver\'s command; un-profiled; the whole line: profiles/r6_bench_default.json)')
r = line['roofline']
P('  headline  %s' % line['config']['workload'])
P('            %.3e grid-point-steps/s, %.4f ms/step, %d reps, timed %.0f ms; kernel %.3f ms per launch '
  '-> %.1f TFLOP/s fp32 = %.1f %% of 157.3 (from the wall clock: %.1f %%); HBM %.3f GB/s algorithmic '
  '(%.4f %% of 8 TB/s)' % (
      line['value'], line['ms_per_step'], line['reps'], line['config']['timed_wall_ms'],
      r['kernel_ms_per_launch'], r['achieved'], 100 * r['frac'], 100 * r['frac_wall'], r['hbm_gbps'],
      100 * r['hbm_frac']))
if os.path.exists(O + '/bench_driver_s20.json'):
  d = json.load(open(O + '/bench_driver_s20.json'))
  shutil.copy(O + '/bench_driver_s20.json', 'profiles/r6_bench_driver_s20.json')
  P('  the driver\'s exact command (python bench.py --steps 20 --warmup 5): %.3e grid-point-steps/s, %d reps, '
    'frac %.1f %%, frac_wall %.1f %%' % (d['value'], d['reps'], 100 * d['roofline']['frac'],
                                      100 * d['roofline']['frac_wall']))
s2 = line['secondary']
P('  secondary %s: %.3e grid-point-steps/s, %.1f TFLOP/s = %.1f %%' % (
    s2['workload'], s2['value'], s2['fp32_tflops'], 100 * s2['frac']))
for name, c in line['configs'].items():
  if name == 'rk_substep_external':
    for b in ('b4096', 'b8192'):
      e = c[b]
      launches = e.get('c_loop_command_ring')
      P('  %-20s %-6s C loop chained %.1f %% (%.1f us per call%s) | Python chained %.1f %% | %sPython unchained %.1f %% | '
        'bit-identical %s' % (name, b, 100 * e['c_loop_chained']['frac'],
                              e['c_loop_chained']['us_per_substep_call'],
                              '',
                              100 * e['python_loop_chained']['frac'],
                              'C loop on the command ring %.1f %% (%d persistent launches for %d calls) | ' % (
                                  100 * launches['frac'], e['ring']['persistent_kernel_launches'],
                                  e['ring']['commands']) if launches else '',
                              100 * e['python_loop_unchained']['frac'],
                              e['drivers_bit_identical'] and e['equals_ddd_integrate_fixed_per_substep']))
  elif 'achieved' in c:
    P('  %-24s %-16s %.3e %s; %.1f %s = %.1f %% of %.1f; %.4f ms per launch%s' % (
        name, c['kernel'], c['value'], c['unit'], c['achieved'], c['roofline_unit'],
        100 * c['frac'], c['peak'], c['kernel_ms_per_launch'],
        ''.join('; %s %.3f' % (k, c[k]) for k in ('frac_algorithmic', 'frac_of_copy_rate') if k in c)))
  elif 'evaluations_per_s' not in c:
    P('  %-24s %-16s %.3e %s' % (name, c.get('kernel', ''), c.get('value', float('nan')), c.get('unit', '')))
  else:
    P('  %-20s %-13s %.1f %s (%.0f evaluations/s); reference: %s ms per evaluation' % (
        name, c['kernel'], c['value'], c['unit'], c['evaluations_per_s'],
        c['reference_ms_per_evaluation']))
cb = line['cpu_baseline']
P('  cpu_baseline (%s, %d threads): %.3e grid-point-steps/s; reference-style (1 core, SciPy RK23 + NumPy): %.3e'
  % (cb['kind'], cb['cores'], cb['value'], cb['reference_style']['value']))
P('  clocks during the timed region: %s\n' % json.dumps(line.get('clocks')))

stats('stats_default', 'python bench.py --cpu-seconds 0   (every kernel of the default line)',
      rows_shown=12)
P('')
for tag, title, keep in (
    ('stats_B4096', 'bench.py --configs none --secondary-batch 0 --cpu-seconds 0 --warmup 1000   (headline alone)',
     ['integrate_kernel']),
    ('stats_B1024', 'same with --batch 1024   (BASELINE configs[1])', ['integrate_kernel']),
    ('stats_B256', 'same with --batch 256   (FOUR 16-row wavefronts per sample, 16x16x4 MFMAs: integrate_kernel<64, 16, ...>)',
     ['integrate_kernel']),
    ('stats_B512', 'same with --batch 512   (four 16-row wavefronts per sample, two workgroups per CU)', ['integrate_kernel']),
    ('stats_weno', 'python profiles/tools/weno_exact_bench.py   (WENO5 exact solver: weno::adaptive_kernel vs generic::adaptive_kernel)',
     ['adaptive_kernel']),
    ('stats_persub', 'bench.py ... --launch-mode per_substep --steps 200 --warmup 200   (two half-ensemble launches per substep)',
     ['substep_multi']),
    ('stats_k7', 'bench.py ... --hparams {"kernel_size": 7} --steps 200   (Tower<7, 1>, streamed weights)',
     ['integrate_kernel']),
    ('stats_f64', 'bench.py ... --hparams {"filter_size": 64} --steps 200   (Tower<5, 2>, streamed weights)',
     ['integrate_kernel']),
    ('stats_stream_step', 'bench.py ... --equation kdv --baseline-stencils --launch-mode per_step --batch 262144   (fused step, fixed stencils)',
     ['fixed_step_kernel']),
    ('stats_lean', 'bench.py ... --hparams {"num_layers": 1} --warmup 1000   (one-layer net: lean::integrate_kernel)',
     ['lean::integrate_kernel']),
    ('stats_adaptive', 'bench.py --configs adaptive_rk23,adaptive_ks_n256_b1024,rk_substep_external   (adaptive kernels; external-driver substeps)',
     ['adaptive_kernel', 'substep_multi'])):
  stats(tag, title, rows_shown=4, keep=keep)
  P('')
P('(bench.py times the R launches of a timed region with ONE HIP-event pair on the launch stream; for the persistent')
P(' kernels its per-launch figure is the rocprofv3 average + the dependent-launch gap.  In the per-substep modes two')
P(' launches run side by side on two streams: a launch\'s own duration is about twice the per-substep figure.)\n')

P('## PMC (rocprofv3 --kernel-trace --pmc ...): bench.py --configs none --secondary-batch 0 --steps 1000 --warmup 0 --preheat-ms 0 --min-timed-ms 0')
acc, last = pmc('pmc_sq_B4096', 'integrate_kernel')
if last is not None:
  evals = 4096 * 2000
  P('## B=4096: 4096 wavefronts x 2000 evaluations; grid %s, workgroup %s, LDS %s B, scratch %s, VGPRs %s' % (
      last['Grid_Size'], last['Workgroup_Size'], last['LDS_Block_Size'], last['Scratch_Size'],
      last['VGPR_Count']))
  for name, values in sorted(acc.items()):
    value = min(v[0] for v in values)
    quad = name in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU')
    P('  %-26s total %.4g   per wave-evaluation %8.0f %s' % (
        name, value, value * (4 if quad else 1) / evals,
        'cycles (quad-cycles x4)' if quad else 'instructions' if 'INSTS' in name else 'cycles'))
acc, last = pmc('pmc_sq_lean', 'lean::integrate_kernel')
if last is not None:
  evals = 4096 * 2000
  P('## one-layer net on lean::integrate_kernel, B=4096: 4096 wavefronts x 2000 evaluations; workgroup %s, LDS %s B, '
    'scratch %s, VGPRs %s' % (last['Workgroup_Size'], last['LDS_Block_Size'], last['Scratch_Size'], last['VGPR_Count']))
  for name, values in sorted(acc.items()):
    value = min(v[0] for v in values)
    quad = name in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU')
    P('  %-26s total %.4g   per wave-evaluation %8.0f %s' % (
        name, value, value * (4 if quad else 1) / evals,
        'cycles (quad-cycles x4)' if quad else 'instructions' if 'INSTS' in name else 'cycles'))
acc, _ = pmc('pmc_clk_B4096', 'integrate_kernel')
if acc['GRBM_GUI_ACTIVE']:
  best = min(acc['GRBM_GUI_ACTIVE'], key=lambda v: v[1])
  mhz = best[0] / 8 / best[1] * 1e3
  cycles = best[1] * 1e-9 * mhz * 1e6 / (4 * 2000)
  P('  GRBM_GUI_ACTIVE %.4g (sum over 8 XCDs) over %.3f ms -> %.0f MHz; %.0f shader cycles per wave-evaluation slot '
    'of every SIMD; matrix pipe busy %.1f %% (14 616 MFMA cycles per evaluation)' % (
        best[0], best[1] / 1e6, mhz, cycles, 100 * 14616 / cycles))
if os.path.exists(O + '/weno_exact.txt'):
  shutil.copy(O + '/weno_exact.txt', 'profiles/r6_weno_exact.txt')
P('')

P('## HBM traffic (FETCH_SIZE and WRITE_SIZE in SEPARATE runs; KB per dispatch; first = first dispatch of the process)')
traffic = {}
patterns = {'persub': 'substep_multi_kernel', 'perstep': 'step_multi_kernel', 'stream': 'fixed_substep',
            'stream_step': 'fixed_step_kernel', 'lean': 'lean::integrate_kernel'}
for tag in ('B4096', 'B1024', 'B256', 'kdv', 'ks256', 'perstep', 'stream_step', 'lean'):
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    acc, _ = pmc('pmc_%s_%s' % (counter, tag), patterns.get(tag, 'integrate_kernel'))
    v = [x[0] for x in acc[counter]]
    if not v:
      P('  %-8s %-10s (no data)' % (tag, counter))
      continue
    traffic[(tag, counter)] = sum(v) / len(v)
    P('  %-8s %-10s dispatches %4d  mean %10.1f KB  first %10.1f  min %10.1f  max %10.1f' % (
        tag, counter, len(v), sum(v) / len(v), v[0], min(v), max(v)))
P('  Persistent integrators: one launch reads y0, the per-sample forcing rows (float4 x 20 modes = 320 B per sample,')
P('    Burgers only), the MFMA-packed weights once per XCD L2 and writes one snapshot.  FETCH_SIZE tallies 64 B per')
P('    fabric request although a wavefront\'s coalesced 256-B row goes out as 128-B requests: it reports HALF the bytes')
P('    (MI355X_MICROARCH.md, HBM section): x2 applied in profiles/r6_hbm_traffic.json.  k7 / f64: the towers with')
P('    streamed weights re-read their layers from L2 every evaluation -- FETCH_SIZE shows what of that reaches HBM.')
P('  stream / stream_step: fixed stencils, B=262144 x N=64 = 64 MiB per array; per substep (20 B per point and step)')
P('    and fused step (8 B per point and step).\n')

open('profiles/r6_rocprof_summary.txt', 'w').write('\n'.join(out) + '\n')

entries = []


def entry(tag, match, correction, dispatches, note, command):
  if (tag, 'FETCH_SIZE') not in traffic or (tag, 'WRITE_SIZE') not in traffic:
    return
  f, w = traffic[(tag, 'FETCH_SIZE')], traffic[(tag, 'WRITE_SIZE')]
  entries.append(dict(match=match, fetch_size_kb=dispatches * f, write_size_kb=dispatches * w,
                      fetch_correction=correction,
                      traffic_bytes_per_launch=int(round(1024 * dispatches * (correction * f + w))),
                      note=note, command=command))


cmd = ('rocprofv3 --kernel-trace --pmc {FETCH_SIZE | WRITE_SIZE} --output-format csv -- python bench.py '
       '--configs none --secondary-batch 0 --cpu-seconds 0 --warmup 0 --preheat-ms 0 --min-timed-ms 0 ')
base = dict(equation='ConservativeBurgersEquation', num_points=64, fixed=False)
half = ('coalesced 256-B wavefront rows / 16-byte-per-lane table reads: FETCH_SIZE reports half the bytes on '
        'gfx950, x2 applied')
entry('B4096', dict(base, batch_per_gpu=4096, launch_mode='persistent'), 2.0, 1,
      'persistent integrator, any --steps (one snapshot); ' + half, cmd + '--steps 1000')
entry('B1024', dict(base, batch_per_gpu=1024, launch_mode='persistent'), 2.0, 1,
      'persistent integrator, any --steps (one snapshot); ' + half, cmd + '--batch 1024 --steps 1000')
entry('B256', dict(base, batch_per_gpu=256, launch_mode='persistent'), 2.0, 1,
      'persistent integrator on four 16-row wavefronts per sample (kQuad); ' + half, cmd + '--batch 256 --steps 1000')
entry('kdv', dict(equation='ConservativeKdVEquation', num_points=64, fixed=False, batch_per_gpu=4096,
                  launch_mode='persistent'), 2.0, 1, 'persistent integrator; ' + half,
      cmd + '--equation kdv --steps 1000')
entry('ks256', dict(equation='ConservativeKSEquation', num_points=256, fixed=False, batch_per_gpu=8192,
                    launch_mode='persistent'), 2.0, 1, 'persistent integrator, 256-row groups; ' + half,
      cmd + '--equation ks --num-points 256 --batch 8192 --steps 400')
entry('persub', dict(base, batch_per_gpu=4096, launch_mode='per_substep'), 2.0, 2,
      'per substep = two half-ensemble dispatches of substep_multi_kernel, mean over the two midpoint stages; ' + half,
      cmd + '--launch-mode per_substep --steps 20')
entry('perstep', dict(base, batch_per_gpu=4096, launch_mode='per_step'), 2.0, 2,
      'per step = two half-ensemble dispatches of step_multi_kernel; ' + half,
      cmd + '--launch-mode per_step --steps 20')
entry('stream', dict(equation='ConservativeKdVEquation', num_points=64, batch_per_gpu=262144,
                     launch_mode='per_substep', fixed=True), 2.0, 1,
      'stream_fixed kernel, mean over the two midpoint stages; 16-byte-per-lane loads: FETCH_SIZE '
      'reports half the bytes on gfx950, x2 applied',
      cmd + '--equation kdv --baseline-stencils --launch-mode per_substep --batch 262144 --steps 20')
entry('stream_step', dict(equation='ConservativeKdVEquation', num_points=64, batch_per_gpu=262144,
                          launch_mode='per_step', fixed=True), 2.0, 1,
      'fused-step streaming kernel (all midpoint stages in one launch); 16-byte-per-lane loads: x2 applied',
      cmd + '--equation kdv --baseline-stencils --launch-mode per_step --batch 262144 --steps 20')
entry('lean', dict(base, batch_per_gpu=4096, launch_mode='persistent', hparams={'num_layers': 1}), 2.0, 1,
      'lean::integrate_kernel (one-layer net, persistent): y0 + forcing rows + cos / sin table in, one snapshot out; ' + half,
      cmd + "--hparams '{\"num_layers\": 1}' --steps 1000")
entry('k7', dict(base, batch_per_gpu=4096, launch_mode='persistent', hparams={'kernel_size': 7}), 2.0, 1,
      'Tower<7, 1>: layer weights streamed from L2 every evaluation (what reaches HBM is the first touch per XCD); '
      'WRITE_SIZE includes the 64-byte scratch frame of this kernel; ' + half,
      cmd + "--hparams '{\"kernel_size\": 7}' --steps 200")
entry('f64', dict(base, batch_per_gpu=4096, launch_mode='persistent', hparams={'filter_size': 64}), 2.0, 1,
      'Tower<5, 2>: layer weights streamed from L2 every evaluation; ' + half,
      cmd + "--hparams '{\"filter_size\": 64}' --steps 200")
json.dump(dict(source='gpurun_out/r6prof/pmc_{FETCH,WRITE}_SIZE_* (round 6, profiles/tools/collect_r6.sh), '
                      'summarized in profiles/r6_rocprof_summary.txt', entries=entries),
          open('profiles/r6_hbm_traffic.json', 'w'), indent=1)
print('\n'.join(out))
