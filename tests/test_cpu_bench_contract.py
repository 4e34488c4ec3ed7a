"""bench.py helpers that do not need a GPU: the measured-traffic lookup and the
committed PMC tables it reads (profiles/r*_hbm_traffic.json, newest first)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
  spec = importlib.util.spec_from_file_location('bench', os.path.join(ROOT, 'bench.py'))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def test_measured_traffic_lookup():
  bench = _bench()
  tables = [json.load(open(os.path.join(ROOT, 'profiles', name)))
            for name in bench.TRAFFIC_TABLES]
  newest = tables[0]
  assert newest['entries'], 'PMC table is empty'
  # the default run (north_star target: batch 4096) and configs[1] are profiled
  for batch in (4096, 1024):
    got, source, _ = bench.measured_traffic('ConservativeBurgersEquation', 64, batch,
                                            'persistent', False)
    assert got and source == 'profiles/' + bench.TRAFFIC_TABLES[0]
  for entry in newest['entries']:
    m = entry['match']
    got, source, _ = bench.measured_traffic(m['equation'], m['num_points'],
                                            m['batch_per_gpu'], m['launch_mode'], m['fixed'],
                                            hparams=m.get('hparams'))   # (other nets: own entries)
    assert got == entry['traffic_bytes_per_launch'] and source.startswith('profiles/')
    # FETCH_SIZE (with the gfx950 correction) + WRITE_SIZE, in bytes
    want = 1024 * (entry['fetch_correction'] * entry['fetch_size_kb'] + entry['write_size_kb'])
    assert abs(got - want) < 1.0
  # the headline configuration is in the table; unprofiled ones report null
  assert bench.measured_traffic('ConservativeBurgersEquation', 64, 1024, 'persistent', False)[0]
  assert bench.measured_traffic('ConservativeBurgersEquation', 64, 1000, 'persistent',
                                False) == (None, None, None)


def test_bench_defaults_match_baseline_config():
  """Primary: BASELINE.json north_star target (Burgers N=64, batch 4096, 1 GPU);
  secondary in the same run: configs[1] (batch 1024), 1000 steps."""
  bench = _bench()
  args = bench.parse_args([])
  assert (args.gpus, args.steps, args.batch, args.num_points, args.equation) == (
      1, 1000, 4096, 64, 'burgers')
  assert args.secondary_batch == 1024
  # a timed region the driver's GPU-activity sampler can see
  assert args.preheat_ms >= 200.0 and args.min_timed_ms >= 1000.0
  # the rest of the contract rides in the same run at N = 1
  assert args.configs == 'all' and set(bench.CONFIG_NAMES) >= {
      'kdv_n64_b4096', 'ks_n256_b8192', 'burgers_per_substep', 'burgers_per_step', 'stream_fixed',
      'differentiator_b1', 'adaptive_rk23'}
  # BASELINE configs[4]: 65 536 samples over 8 GPUs = 8 192 per GPU
  assert bench.parse_args(['--gpus', '8']).batch == 8192
  assert bench.parse_args(['--gpus', '8', '--batch', '512']).batch == 512
  assert args.scheme == 'midpoint' and args.launch_mode == 'persistent'
  assert bench.PEAK_FP32_TFLOPS == 157.3 and bench.PEAK_HBM_GBPS == 8000.0
