"""Token statistics of the paired persistent kernel (traced instantiation)."""
import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddd1d_amd
from ddd1d_amd import equations, model as model_lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
for item in sys.argv[3:]:
    name, _, value = item.partition('=')
    ddd1d_amd._lib.debug_set_option(name, int(value))
hp = ddd1d_amd.create_hparams('burgers', conservative=True, resample_factor=8, equation_kwargs=json.dumps({'num_points': 512}))
_, eq = equations.from_hparams(hp)
m = model_lib.LearnedStencilModel(eq, hp)
m.set_forcing(model_lib.batched_forcing_parameters(range(B), nparams=20))
y0 = torch.randn(B, 64, device='cuda') * 0.3
m.integrate_fixed(y0, steps, dt=1e-3, save_every=steps)   # warm
trace = torch.zeros(B * 8, dtype=torch.int64, device='cuda')
ddd1d_amd._lib.debug_set_option('trace_ptr', trace.data_ptr())
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
m.integrate_fixed(y0, steps, dt=1e-3, save_every=steps)
e1.record()
torch.cuda.synchronize()
ddd1d_amd._lib.debug_set_option('trace_ptr', 0)
print('kernel', m.kernel_name, 'ms', e0.elapsed_time(e1), 'us/step', e0.elapsed_time(e1) * 1e3 / steps)
tr = trace.cpu().numpy().reshape(B, 8)
info = tr[:, 0]
simd, count, partner, w = info & 0xf, (info >> 4) & 0xf, (info >> 8) & 0xf, (info >> 12) & 0xf
hw = tr[:, 1]
print('count histogram', np.bincount(count.astype(int)))
print('first workgroup: w, simd, partner, hw simd, hw wave, hw cu:')
for g in range(8):
    print('  ', w[g], simd[g], partner[g], (hw[g] >> 4) & 3, hw[g] & 15, (hw[g] >> 8) & 15, 'xcc', tr[g, 7])
total = tr[:, 6].astype(np.float64)
print('cycles per wave: mean %.0f; per eval %.0f' % (total.mean(), total.mean() / (2 * steps)))
print('waited fraction: mean %.3f min %.3f max %.3f' % ((tr[:, 2] / total).mean(), (tr[:, 2] / total).min(), (tr[:, 2] / total).max()))
print('polls per acquire: %.2f; acquires per eval %.2f; timeouts %d' % (tr[:, 3].sum() / max(tr[:, 4].sum(), 1), tr[:, 4].mean() / (2 * steps), tr[:, 5].sum()))
print('wait cycles per acquire: %.0f' % (tr[:, 2].sum() / max(tr[:, 4].sum(), 1)))
