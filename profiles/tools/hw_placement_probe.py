import ctypes, os, sys, numpy as np, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ddd1d_amd
lib = ddd1d_amd._lib.load_probe_library()   # libddd1d_probe.so (__graft_entry__.build_probe)
torch.zeros(1).cuda()
blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
out = np.zeros((blocks, 4), dtype=np.uint32)
lib.ddd_debug_hwid.restype = ctypes.c_int
rc = lib.ddd_debug_hwid(out.ctypes.data_as(ctypes.c_void_p), blocks, 200000)
print('rc', rc)
hw = out[:, 0]; xcc = out[:, 1] & 0xf
wave = hw & 0xf; simd = (hw >> 4) & 3; pipe = (hw >> 6) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print('wave ids', collections.Counter(wave.tolist()))
print('simd', collections.Counter(simd.tolist()))
key = collections.defaultdict(list)
for b in range(blocks):
    key[(int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b]), int(simd[b]))].append((b, int(wave[b])))
sizes = collections.Counter(len(v) for v in key.values())
print('waves per (xcc,se,sh,cu,simd):', sizes, 'distinct SIMDs', len(key))
for k in list(key)[:12]:
    print(k, key[k])
per_cu = collections.Counter((int(xcc[b]), int(se[b]), int(sh[b]), int(cu[b])) for b in range(blocks))
print('workgroups per CU:', collections.Counter(per_cu.values()), 'distinct CUs', len(per_cu))
