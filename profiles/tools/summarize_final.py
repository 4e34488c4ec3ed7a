"""Turn gpurun_out/final (profiles/tools/collect_final.sh) into the committed
profiles/r1_final_* files.  Run in the authoring container after the gpurun call."""
import collections
import csv
import glob
import json
import os
import shutil

O = 'gpurun_out/final'
out = []
P = out.append


def stats(tag, title):
  P('## rocprofv3 --kernel-trace --stats --output-format csv -- ' + title)
  P('%-96s %6s %14s %13s %8s %12s %12s' % ('kernel', 'calls', 'total_ns', 'avg_ns', 'pct',
                                            'min_ns', 'max_ns'))
  rows = list(csv.DictReader(open(O + '/prof_%s/bench_kernel_stats.csv' % tag)))
  for r in rows[:6]:
    P('%-96s %6s %14s %13.0f %8s %12s %12s' % (r['Name'][:96], r['Calls'], r['TotalDurationNs'],
                                               float(r['AverageNs']), r['Percentage'],
                                               r['MinNs'], r['MaxNs']))


def pmc(tag, pattern):
  acc = collections.defaultdict(float)
  last, dur = None, 0
  for r in csv.DictReader(open(O + '/%s/pmc_counter_collection.csv' % tag)):
    if pattern in r['Kernel_Name']:
      acc[r['Counter_Name']] += float(r['Counter_Value'])
      last = r
  for r in csv.DictReader(open(O + '/%s/pmc_kernel_trace.csv' % tag)):
    if pattern in r['Kernel_Name']:
      dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
  return acc, last, dur


P('# Round 1 -- final measurement run (profiles/tools/collect_final.sh, one gpurun call, 1 x MI355X)')
P('# raw outputs: gpurun_out/final (scratch); this file: profiles/tools/summarize_final.py\n')
P(open(O + '/smoke.txt').read().strip())
P(open(O + '/pytest_gpu.txt').read().strip() + '   (pytest tests -m gpu)\n')

stats('default', 'python bench.py --cpu-seconds 0   (Burgers N=64, B=1024, 1000 midpoint steps, persistent)')
d = json.load(open(O + '/bench_default.json'))
P('(integrate_kernel<64,64,float,true,1,false> = the ConservativeBurgers specialisation; two calls = 100-step')
P(' warm-up launch [MinNs] + the timed 1000-step launch [MaxNs]; bench.py HIP-event time of the timed launch in a')
P(' separate un-profiled run: %.3f ms -> %.2f TFLOP/s fp32 = %.1f %% of 157.3)\n' % (
    d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved'], 100 * d['roofline']['frac']))
if os.path.exists(O + '/prof_default_w1000/bench_kernel_stats.csv'):
  stats('default_w1000', 'python bench.py --warmup 1000 --cpu-seconds 0   (warm-up launch as long as the timed one)')
  dd = json.loads([l for l in open(O + '/prof_default_w1000.log') if l.startswith('{')][0])
  P('(the first launch of the process is cold -- code object load, clocks ramping -- and runs ~1.7x slower; the second,')
  P(' timed launch [MinNs] is the steady state and agrees with bench.py\'s HIP-event time in the same run: %.3f ms)\n' % (
      dd['roofline']['kernel_ms_per_launch']))
stats('stream', 'python bench.py --equation kdv --baseline-stencils --launch-mode per_substep '
                '--batch 262144 --steps 100 --warmup 10 --cpu-seconds 0')
d = json.load(open(O + '/bench_fixed_kdv_persub.json'))
P('(stream::fixed_substep_kernel, 220 launches; bench.py: %.4f ms per launch, %.0f GB/s algorithmic = %.1f %% of 8 TB/s)\n' % (
    d['roofline']['kernel_ms_per_launch'], d['roofline']['achieved'], 100 * d['roofline']['frac']))

cycles = {}
for tag, batch in (('pmc_sq_B4096', 4096), ('pmc_sq_B1024', 1024)):
  acc, last, dur = pmc(tag, 'integrate_kernel')
  P('## PMC (rocprofv3 --kernel-trace --pmc SQ_*): bench.py --batch %d --steps 200 --warmup 0 --cpu-seconds 0' % batch)
  P('## %d wavefronts x 400 RHS evaluations; kernel %.3f ms under the profiler; grid %s, workgroup %s, LDS %s B, scratch %s' % (
      batch, dur / 1e6, last['Grid_Size'], last['Workgroup_Size'], last['LDS_Block_Size'],
      last['Scratch_Size']))
  evals = batch * 400
  for name, value in sorted(acc.items()):
    quad = name not in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_BUSY_CYCLES')
    unit = ('cycles (quad-cycles x4)' if quad else
            'instructions' if 'INSTS' in name else 'cycles')
    P('  %-26s total %.4g   per wave-evaluation %8.0f %s' % (name, value,
                                                           value * (4 if quad else 1) / evals, unit))
  wave = acc['SQ_WAVE_CYCLES'] * 4 / evals
  cycles[batch] = wave
  if batch == 4096:
    step_us = 1e3 * json.load(open(O + '/bench_B4096.json'))['ms_per_step']
    clock = 4 * wave / step_us / 1e3
    P('  two wavefronts share a SIMD: matrix pipe busy = 2 x 16000 / %.0f = %.1f %% of the SIMD\'s cycles.' % (
        wave, 100 * 32000 / wave))
    P('  Cross-check with the un-profiled rate (%.1f us per step = 8 wave-evaluations per SIMD, two at a time):' % step_us)
    P('  4 x %.0f cycles / %.1f us = %.2f GHz sustained shader clock under this load (peak 2.4 GHz).\n' % (
        wave, step_us, clock))
  else:
    step_us = 1e3 * json.load(open(O + '/bench_default.json'))['ms_per_step']
    P('  one wavefront per SIMD: matrix pipe busy = 16000 / %.0f = %.1f %% of the wave\'s cycles under the profiler;' % (
        wave, 100 * 16000 / wave))
    P('  un-profiled: %.2f us per step = 2 evaluations per SIMD = %.1f k cycles per evaluation at 2.4 GHz.' % (
        step_us, step_us * 2.4 / 2))
    P('  (this short cold launch ran at ramping clocks: its profiled duration is not the bench time)\n')

P('## HBM traffic (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE passes; KB per dispatch)')
traffic = {}
for tag, pattern in (('default', 'integrate_kernel'), ('stream', 'fixed_substep')):
  for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = [float(r['Counter_Value'])
            for r in csv.DictReader(open(O + '/pmc_%s_%s/pmc_counter_collection.csv' % (counter, tag)))
            if r['Counter_Name'] == counter and pattern in r['Kernel_Name']]
    traffic[(tag, counter)] = sum(vals) / len(vals)
    P('  %-8s %-11s dispatches %3d  mean %10.1f KB  min %10.1f  max %10.1f' % (
        tag, counter, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
P('  default (persistent integrator, B=1024): read + written per launch against ~20 ms of compute; algorithmic')
P('    256 KiB (y0) + 320 KiB (forcing rows) + ~32 KiB (weights) read, 256 KiB written: no re-reads.')
P('  stream (fixed stencils, B=262144 x N=64 = 64 MiB per array): WRITE_SIZE = 65536 KB exactly; FETCH_SIZE alternates')
P('    ~32768 / ~65536 KB between the two midpoint stages = HALF of the 64 / 128 MiB the float4 loads fetch (gfx950')
P('    counter reports 1/2 for 16-byte-per-lane streams, MI355X_MICROARCH.md); corrected traffic = algorithmic bytes.\n')

P('## s_memtime phase trace (profiles/tools/trace_phases.py, dedicated traced instantiation; cycles per evaluation)')
P(open(O + '/phases_B1024.txt').read().rstrip())
P(open(O + '/phases_B4096.txt').read().rstrip())
P('  (pure matrix-pipe time: input+hidden 10752, output layer 5248 cycles)\n')

P('## bench.py matrix (JSON lines in profiles/r1_final_bench_*.json)')
for f in sorted(glob.glob(O + '/bench_*.json')):
  d = json.load(open(f))
  r = d['roofline']
  P('  %-28s %.3e gps/s  %-4s %9.2f %-8s frac %.3f  %s' % (
      os.path.basename(f)[6:-5], d['value'], r['bound'], r['achieved'], r['unit'], r['frac'],
      d['config']['kernel']))
  shutil.copy(f, 'profiles/r1_final_' + os.path.basename(f))
d = json.load(open(O + '/bench_default.json'))
P('  cpu_baseline (default run): %s' % json.dumps(d['cpu_baseline']))
open('profiles/r1_final_rocprof_summary.txt', 'w').write('\n'.join(out) + '\n')

table = {
    'source': 'gpurun_out/final/pmc_{FETCH,WRITE}_SIZE_{default,stream} (round 1), summarized in '
              'profiles/r1_final_rocprof_summary.txt',
    'entries': [
        {'match': {'equation': 'ConservativeBurgersEquation', 'num_points': 64,
                   'batch_per_gpu': 1024, 'launch_mode': 'persistent', 'fixed': False},
         'fetch_size_kb': traffic[('default', 'FETCH_SIZE')],
         'write_size_kb': traffic[('default', 'WRITE_SIZE')], 'fetch_correction': 1.0,
         'traffic_bytes_per_launch': int(round(1024 * (traffic[('default', 'FETCH_SIZE')] +
                                                       traffic[('default', 'WRITE_SIZE')]))),
         'note': '4-byte-per-lane loads: FETCH_SIZE matches the byte count (calibrated on the '
                 'per-substep kernel, profiles/r1_rocprof_summary.txt)'},
        {'match': {'equation': 'ConservativeKdVEquation', 'num_points': 64,
                   'batch_per_gpu': 262144, 'launch_mode': 'per_substep', 'fixed': True},
         'fetch_size_kb': traffic[('stream', 'FETCH_SIZE')],
         'write_size_kb': traffic[('stream', 'WRITE_SIZE')], 'fetch_correction': 2.0,
         'traffic_bytes_per_launch': int(round(1024 * (2 * traffic[('stream', 'FETCH_SIZE')] +
                                                       traffic[('stream', 'WRITE_SIZE')]))),
         'note': 'stream_fixed kernel, mean over the two midpoint stages; 16-byte-per-lane loads: '
                 'FETCH_SIZE reports half the bytes on gfx950, x2 applied'},
    ]}
json.dump(table, open('profiles/r1_hbm_traffic.json', 'w'), indent=1)
print('\n'.join(out[-40:]))
