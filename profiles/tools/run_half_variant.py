import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import ddd1d_amd
lib = ddd1d_amd._lib.load_library(os.path.join(ROOT, 'data-driven-discretization-1d_amd/csrc/libddd1d_%s.so' % sys.argv[1]))
import test_gpu_half_tower as t
cases = [('burgers', True, 64, dict(filter_size=16)), ('burgers', False, 64, dict(filter_size=12, kernel_size=3)),
         ('kdv', True, 64, dict(filter_size=16)), ('kdv', False, 64, dict(filter_size=8)),
         ('ks', True, 64, dict(filter_size=16, kernel_size=4)), ('ks', False, 64, dict(filter_size=5)),
         ('burgers', True, 32, dict(filter_size=16)), ('kdv', True, 16, dict(filter_size=10))]
for c in cases:
  t.test_block_diagonal_tower(*c)
  print('ok', c, flush=True)
t.test_block_diagonal_tower_nan_mask()
print('nan ok')
