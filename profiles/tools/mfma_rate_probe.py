import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ddd1d_amd
lib = ddd1d_amd._lib.load_library()
torch.zeros(1).cuda()
lib.ddd_debug_mfma_rate.argtypes = [ctypes.c_int]*4 + [ctypes.POINTER(ctypes.c_double)]*2
for blocks in (1024, 2048):
  for is32 in (1, 0):
    for chains in (1, 2, 4):
      t = ctypes.c_double(); w = ctypes.c_double()
      rc = lib.ddd_debug_mfma_rate(chains, is32, blocks, 20000, ctypes.byref(t), ctypes.byref(w))
      waves_per_simd = blocks / 1024
      print('blocks %d %s chains %d: %.1f ticks/MFMA per wave, wall %.2f ns per MFMA per wave -> %.2f ns per SIMD-MFMA' % (
          blocks, '32x32x2' if is32 else '16x16x4', chains, t.value, w.value, w.value / waves_per_simd))
