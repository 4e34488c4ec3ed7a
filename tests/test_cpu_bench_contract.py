"""bench.py helpers that do not need a GPU: the measured-traffic lookup and the
committed PMC table it reads (profiles/r1_hbm_traffic.json)."""
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
  table = json.load(open(os.path.join(ROOT, 'profiles', 'r1_hbm_traffic.json')))
  assert table['entries'], 'PMC table is empty'
  for entry in table['entries']:
    m = entry['match']
    got = bench.measured_traffic(m['equation'], m['num_points'], m['batch_per_gpu'],
                                 m['launch_mode'], m['fixed'])
    assert got == entry['traffic_bytes_per_launch']
    # FETCH_SIZE (with the gfx950 correction) + WRITE_SIZE, in bytes
    want = 1024 * (entry['fetch_correction'] * entry['fetch_size_kb'] + entry['write_size_kb'])
    assert abs(got - want) < 1.0
  # the headline configuration is in the table; unprofiled ones report null
  assert bench.measured_traffic('ConservativeBurgersEquation', 64, 1024, 'persistent', False)
  assert bench.measured_traffic('ConservativeBurgersEquation', 64, 1000, 'persistent', False) is None


def test_bench_defaults_match_baseline_config():
  """BASELINE.json configs[1]: Burgers N=64, batch 1024, 1000 steps."""
  import sys
  bench = _bench()
  argv, sys.argv = sys.argv, ['bench.py']
  try:
    args = bench.parse_args()
  finally:
    sys.argv = argv
  assert (args.gpus, args.steps, args.batch, args.num_points, args.equation) == (
      1, 1000, 1024, 64, 'burgers')
  assert args.scheme == 'midpoint' and args.launch_mode == 'persistent'
  assert bench.PEAK_FP32_TFLOPS == 157.3 and bench.PEAK_HBM_GBPS == 8000.0
