"""Wave lifetimes of the one-launch-per-substep mode (probe library): where the fixed
cost per launch goes.  Every wavefront of every substep launch stamps s_memrealtime
(100 MHz) at its first instruction, when its weights / first forcing sums are ready,
after each row group and at its last instruction, plus its hardware id.
  python profiles/tools/substep_wave_trace.py [batch] [steps]"""
import collections, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ddd1d_amd
from ddd1d_amd import equations, model as model_lib
ddd1d_amd._lib.load_probe_library()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ROWS = 2048
hp = ddd1d_amd.create_hparams('burgers', conservative=True, resample_factor=8,
                              equation_kwargs=json.dumps({'num_points': 512}))
_, eq = equations.from_hparams(hp)
m = model_lib.LearnedStencilModel(eq, hp)
m.set_forcing(model_lib.batched_forcing_parameters(range(B), nparams=20))
y0 = torch.randn(B, 64, device='cuda') * 0.3
m.integrate_fixed(y0, 400, dt=1e-3, save_every=400, launch_mode='per_substep')   # warm
torch.cuda.synchronize()
launches = steps * 2 * 2   # stages x half-ensembles
trace = torch.zeros(launches * ROWS * 8, dtype=torch.int64, device='cuda')
ddd1d_amd._lib.debug_set_option('walk_trace_ptr', trace.data_ptr())
m.integrate_fixed(y0, steps, dt=1e-3, save_every=steps, launch_mode='per_substep')
torch.cuda.synchronize()
ddd1d_amd._lib.debug_set_option('walk_trace_ptr', 0)
T = trace.cpu().numpy().reshape(launches, ROWS, 8)
used = T[:, :, 0] != 0
print('batch', B, 'steps', steps, 'launches', launches, 'waves per launch', collections.Counter(used.sum(1).tolist()))
t0 = T[:, :, 0][used].min()
us = lambda x: (x.astype(np.float64) - t0) / 100.0

def simd_key(h):
    hw = h & 0xffffffff; xcc = (h >> 32) & 0xf
    return (xcc << 16) | (hw & 0xfff0)    # xcc, se, sh, cu, pipe, simd (wave slot dropped)

rows = []
for l in range(4, launches):            # skip the first step (chains still staggering in)
    u = used[l]
    s0, s1, e_last, end = us(T[l, u, 0]), us(T[l, u, 1]), None, us(T[l, u, 7])
    g1, g2 = us(T[l, u, 2]), us(T[l, u, 3])
    prev = l - 2                         # previous launch of the same chain
    pend = us(T[prev, used[prev], 7])
    keys = simd_key(T[l, u, 6])
    cnt = collections.Counter(keys.tolist())
    same = np.array([cnt[k] for k in keys.tolist()])
    rows.append(dict(
        gap_first=s0.min() - pend.max(),          # predecessor's last wave end -> first wave start
        start_spread=s0.max() - s0.min(),
        math=np.median(us(T[l, u, 4]) - s0), issued=np.median(us(T[l, u, 5]) - s0),
        setup=np.median(s1 - s0), ev1=np.median(g1 - s1), ev2=np.median(g2 - g1),
        life=np.median(end - s0), life_max=(end - s0).max(),
        end_spread=end.max() - np.median(end), span=end.max() - s0.min(),
        pitch=s0.min() - us(T[prev, used[prev], 0]).min(),
        alone=np.median((end - s0)[same == 1]) if (same == 1).any() else np.nan,
        paired=np.median((end - s0)[same == 2]) if (same == 2).any() else np.nan,
        n_paired=int((same == 2).sum()),
    ))
keys = list(rows[0].keys())
print('per launch, median over launches (us):')
for k in keys:
    v = np.array([r[k] for r in rows], dtype=np.float64)
    print('  %-13s median %8.2f   min %8.2f   max %8.2f' % (k, np.nanmedian(v), np.nanmin(v), np.nanmax(v)))
print('''legend: gap_first = last wave of the chain's previous launch ends -> first wave of this launch starts;
  start_spread = first -> last wave start; math / issued = first instruction -> every load of the setup requested and its index math done /
  -> all of them arrived, tables staged; setup = first instruction -> weights + first forcing sums ready;
  ev1 / ev2 = first / second row group; life = wave lifetime; end_spread = median wave end -> last wave end;
  span = first start -> last end; pitch = start-to-start distance of consecutive launches of one chain;
  alone / paired = lifetime of waves that are the only / one of two waves of THEIR launch on a SIMD''')
# one launch in detail: histogram of wave ends relative to the first start
l = launches - 3
u = used[l]
end = us(T[l, u, 7]); s0 = us(T[l, u, 0])
print('launch %d: wave start percentiles (us after first start):' % l,
      np.percentile(s0 - s0.min(), [0, 10, 50, 90, 99, 100]).round(2).tolist())
print('launch %d: wave end percentiles   (us after first start):' % l,
      np.percentile(end - s0.min(), [0, 10, 50, 90, 99, 100]).round(2).tolist())
life = end - s0
print('launch %d: wave lifetime percentiles:' % l, np.percentile(life, [0, 10, 50, 90, 99, 100]).round(2).tolist())
keys = simd_key(T[l, u, 6])
cnt = collections.Counter(keys.tolist())
same = np.array([cnt[k] for k in keys.tolist()])
for n in sorted(set(same.tolist())):
    sel = same == n
    print('  waves on a SIMD holding %d wave(s) of this launch: %d, lifetime median %.2f max %.2f, start median %.2f'
          % (n, sel.sum(), np.median(life[sel]), life[sel].max(), np.median((s0 - s0.min())[sel])))
slow = np.argsort(life)[-8:]
print('  slowest waves: slot, start, setup, ev1, ev2:', [(int(np.nonzero(u)[0][i]), round(float(s0[i] - s0.min()), 2),
      round(float(us(T[l, u, 1])[i] - s0[i]), 2), round(float(us(T[l, u, 2])[i] - us(T[l, u, 1])[i]), 2),
      round(float(us(T[l, u, 3])[i] - us(T[l, u, 2])[i]), 2)) for i in slow])
