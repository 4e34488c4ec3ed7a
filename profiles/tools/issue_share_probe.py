"""Two wavefronts per SIMD: MFMA streamer (wave slot 0) vs VALU / LDS worker (slot 1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ddd1d_amd
lib = ddd1d_amd._lib.load_probe_library()   # libddd1d_probe.so (__graft_entry__.build_probe)
torch.zeros(1).cuda()
fn = lib.ddd_debug_issue_share
fn.argtypes = [ctypes.c_int] * 5 + [ctypes.POINTER(ctypes.c_double)] * 2
MF = {0: 'idle', 1: '32x32x2 x2 chains', 2: '4x4x1 x3 chains', 3: '32x32x2 + 16 nop cycles', 4: '32x32x2 + 32 nop cycles',
      5: '32x32x2 + 48 nop cycles', 6: '32x32x2 + 56 nop cycles', 7: '4x4x1 + s_nop 1', 8: '4x4x1 + s_nop 3'}
WK = {0: 'idle', 1: 'dependent v_fma', 2: '4 independent v_fma', 3: 'LDS read+write chain', 4: 'poll: ds_read+readfirstlane+s_sleep 1'}
def run(mf, work, prio=0):
  a = ctypes.c_double(); b = ctypes.c_double()
  rc = fn(mf, work, 256, 200, prio, ctypes.byref(a), ctypes.byref(b))
  assert rc == 0, rc
  print('mfma %-24s | worker %-38s prio %d | %6.1f ticks per MFMA | %6.1f ticks per worker op' % (MF[mf], WK[work], prio, a.value, b.value))
full = len(sys.argv) > 1
if full:
  for work in (1, 2, 3, 4):
    for mf in (0, 1, 2):
      for prio in ((0, 1) if mf else (0,)):
        run(mf, work, prio)
for mf in (1, 3, 4, 5, 6, 2, 7, 8):
  for work in (0, 1, 2) + ((3,) if mf in (5, 8) else ()):
    run(mf, work)
