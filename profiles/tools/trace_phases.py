import os, sys, json, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddd1d_amd
from ddd1d_amd import equations, model as model_lib
ddd1d_amd._lib.load_probe_library()   # libddd1d_probe.so (__graft_entry__.build_probe)
if 'no_spec' in sys.argv:   # trace the run-time-parameterised kernel instead of the per-equation one
    ddd1d_amd._lib.debug_set_option('no_spec', 1)
    sys.argv.remove('no_spec')
for item in [a for a in sys.argv if a.startswith('ablate=')]:   # skip phases (WRONG results): 1 forcing, 2 projection, 4 output layer, 16 input layer
    ddd1d_amd._lib.debug_set_option('ablate', int(item.split('=')[1]))
    sys.argv.remove(item)
eq_name, extra = 'burgers', {}
for item in [a for a in sys.argv if a.startswith('eq=')]:       # eq=ks: another equation (run-time kernels)
    eq_name = item.split('=')[1]; sys.argv.remove(item)
for item in [a for a in sys.argv if a.startswith('hp=')]:       # hp={"coefficient_grid_min_size": 9}
    extra = json.loads(item[3:]); sys.argv.remove(item)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
hp = ddd1d_amd.create_hparams(eq_name, conservative=True, resample_factor=8, equation_kwargs=json.dumps({'num_points': 512}), **extra)
_, eq = equations.from_hparams(hp)
m = model_lib.LearnedStencilModel(eq, hp)
if eq_name == 'burgers':
    m.set_forcing(model_lib.batched_forcing_parameters(range(B), nparams=20))
y0 = torch.randn(B, 64, device='cuda') * 0.3
trace = torch.zeros(B * 256, dtype=torch.int64, device='cuda')
ddd1d_amd._lib.debug_set_option('trace_ptr', trace.data_ptr())
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
m.integrate_fixed(y0, 2000, dt=1e-3, save_every=2000)   # warm: clocks settled
torch.cuda.synchronize()
raw = trace.cpu().numpy().reshape(B, 256)
ticks, real = raw[:, 254].astype(np.float64), raw[:, 255].astype(np.float64)
print('2000-step launch per wave: s_memtime ticks mean %.0f, s_memrealtime (100 MHz) ticks mean %.0f -> s_memtime rate %.1f MHz' % (ticks.mean(), real.mean(), 100.0 * ticks.mean() / real.mean()))
trace.zero_()
m.integrate_fixed(y0, steps, dt=1e-3, save_every=steps)
torch.cuda.synchronize()
raw = trace.cpu().numpy().reshape(B, 256)
ticks, real = raw[:, 254].astype(np.float64), raw[:, 255].astype(np.float64)
print('whole launch per wave: s_memtime ticks mean %.0f, s_memrealtime (100 MHz) ticks mean %.0f -> s_memtime rate %.1f MHz' % (ticks.mean(), real.mean(), 100.0 * ticks.mean() / real.mean()))
tr = trace.cpu().numpy().reshape(B, 256)[:, :250].reshape(B, 50, 5)
# co-resident pairs (from the placement probe): block b and b + 768 (first round)
for b in (0, 1, 8):
    for partner in (b + 768, b + 1024):
        if partner >= B: continue
        a0 = tr[b, :, 0]; p0 = tr[partner, :, 0]
        print('block', b, 'partner', partner)
        print('  eval period A', np.diff(a0)[:12])
        print('  eval period P', np.diff(p0)[:12])
        print('  start offset P-A per eval', (p0 - a0)[:12])
        print('  A phases (input,hidden,final,epilogue):', (tr[b, 5:9, 1:] - tr[b, 5:9, :4]).tolist())
np.save('gpurun_out/trace.npy', tr)
names = ['u/frc(0-1)', 'input+hidden(1-2)', 'final(2-3)', 'epilogue(3-4)', 'loop(4-0)']
d = np.diff(tr[:, 5:45, :].astype(np.int64), axis=2)            # [B, 40, 4]
loop = tr[:, 6:46, 0].astype(np.int64) - tr[:, 5:45, 4].astype(np.int64)
allp = np.concatenate([d, loop[..., None]], axis=2)
print('B =', B, 'mean cycles per phase over waves/evals:')
for i, n in enumerate(names):
    print('  %-20s mean %8.0f  median %8.0f  min %8.0f' % (n, allp[..., i].mean(), np.median(allp[..., i]), allp[..., i].min()))
print('  period mean', np.diff(tr[:, 5:45, 0].astype(np.int64), axis=1).mean())
