"""Where a command's time goes in the ring kernel (csrc/rhs_ring.h, -DDDD_RING_TRACE=1 variant
build `ringtrace`): s_memrealtime stamps of the first commands of wavefront groups 0..7.
Usage (on the GPU box): python profiles/tools/ring_trace.py [batch]"""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as g
g.build_hip()
path = g.build_hip(variant='ringtrace', variant_flags={'mfma_ring.hip': ['-DDDD_RING_TRACE=1'],
                                                       'capi.hip': ['-DDDD_RING_TRACE=1']})
import bench
import ddd1d_amd
lib = ddd1d_amd._lib.load_library(path)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
args = bench.parse_args([])
a = bench._variant(args)
eq, model, _, y0_host = bench.build_workload(a, 0, batch, unique=min(batch, 4096))
dt = eq.time_step
h = np.float32(dt)
y0 = torch.from_numpy(y0_host).cuda()
y, ystage, ynew = y0.clone(), torch.empty_like(y0), torch.empty_like(y0)
lib.ddd_set_region_mode(model._handle, 2)
steps = 150
for rep in range(2):   # the second region is the one read (stamps overwritten)
  with model.chained_substeps():
    for step in range(steps):
      t = step * dt
      model.rk_substep(t, y, y_base=y, c1=0.5 * h, y_out=ystage)
      model.rk_substep(t + 0.5 * dt, ystage, acc_in=y, c2=h, acc_out=ynew)
      y, ynew = ynew, y
  torch.cuda.synchronize()
ptr = ctypes.c_int64(0); cmds = ctypes.c_int64(0)
lib.ddd_region_stats(model._handle, ctypes.byref(ptr), ctypes.byref(cmds))
stamps = np.ctypeslib.as_array((ctypes.c_uint64 * (8 * 128)).from_address(ptr.value)).reshape(8, 128).astype(np.int64)
gpw = max(1, -(-batch // 2048))
print('batch %d: %d row groups per wavefront; ticks of 10 ns; stamps: 0 command known, 1 cold start done, '
      '2/3 eval0 start/end, 4 stored, 5 barrier, 6/7 eval1 start/end, 8 stored, 9 barrier, 10 counted' % (batch, gpw))
for w in range(3):
  print('group %d' % w)
  for c in range(9):
    row = stamps[w, 12 * c: 12 * c + 11]
    nxt = stamps[w, 12 * (c + 1)]
    if row[0] == 0 or nxt == 0: break
    keys = [k for k in range(11) if row[k] != 0]
    out = []
    for i, k in enumerate(keys):
      end = row[keys[i + 1]] if i + 1 < len(keys) else nxt
      out.append('%d:%d' % (k, end - row[k]))
    print('  cmd %2d: ' % c + ' '.join(out) + '  | command %d' % (nxt - row[0]))
model.close()
