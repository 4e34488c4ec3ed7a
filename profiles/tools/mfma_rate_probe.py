import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ddd1d_amd
lib = ddd1d_amd._lib.load_probe_library()   # libddd1d_probe.so (__graft_entry__.build_probe)
torch.zeros(1).cuda()
lib.ddd_debug_mfma_rate.argtypes = [ctypes.c_int]*4 + [ctypes.POINTER(ctypes.c_double)]*2
NAMES = {1: '32x32x2', 0: '16x16x4', 2: '4x4x1_16b(cbsz4)'}
for blocks in (1024, 2048):
  for kind in (1, 0, 2):
    for chains in ((1, 2, 3, 4) if kind == 2 else (1, 2, 4)):
      t = ctypes.c_double(); w = ctypes.c_double()
      rc = lib.ddd_debug_mfma_rate(chains, kind, blocks, 20000, ctypes.byref(t), ctypes.byref(w))
      assert rc == 0
      waves_per_simd = blocks / 1024
      print('blocks %d %s chains %d: %.1f ticks/MFMA per wave, wall %.2f ns per MFMA per wave -> %.2f ns per SIMD-MFMA' % (
          blocks, NAMES[kind], chains, t.value, w.value, w.value / waves_per_simd))
