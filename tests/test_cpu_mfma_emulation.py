"""Lane-level NumPy emulation of the MFMA kernel's data movement (no GPU).

Transcribes, formula by formula, (a) the weight packing of
csrc/capi.hip::pack_mfma_weights, (b) the operand gathers / result scatters of
csrc/rhs_mfma.h (input_layer, hidden_layer, final_layer4) and (c) the CDNA4
f32 MFMA register layouts the kernels assume
(cdna_hip_programming.md section 3):

  v_mfma_f32_32x32x2_f32 : lane l supplies A[i = l & 31][k = l >> 5] and
      B[k = l >> 5][j = l & 31]; register r of lane l holds
      D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
  v_mfma_f32_4x4x1_16b_f32, cbsz = 4, abid = b : register r of lane l
      accumulates A(lane 4 b + r) * B(lane l).

and checks that the composition equals the oracle's conv tower.  The layouts
themselves are verified on hardware by ddd_selftest_mfma_layout.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import oracle  # noqa: E402

ROWS, HS = 256, 36
LANES = np.arange(64)


def mfma32(a, b, acc):
  """acc[lane, r] += D-layout of A(32x2) @ B(2x32)."""
  A = np.stack([a[:32], a[32:]], axis=1).astype(np.float64)     # [i, k]
  B = np.stack([b[:32], b[32:]], axis=0).astype(np.float64)     # [k, j]
  D = A @ B
  for r in range(16):
    i = (r & 3) + 8 * (r >> 2) + 4 * (LANES >> 5)
    acc[:, r] += D[i, LANES & 31]
  return acc


def pack_input(w, b):          # capi.hip: input layer
  packed = np.zeros((3, 64))
  for s in range(3):
    for lane in range(64):
      k, ch = 2 * s + (lane >> 5), lane & 31
      packed[s, lane] = w[k, 0, ch] if k < 5 else b[ch]
  return packed


def pack_hidden(w, b):         # capi.hip: hidden layer
  packed = np.zeros((81, 64))
  for s in range(80):
    tap, jj = s // 16, s % 16
    for lane in range(64):
      cin, cout = 16 * (lane >> 5) + jj, lane & 31
      packed[s, lane] = w[tap, cin, cout]
  for lane in range(64):
    packed[80, lane] = b[lane & 31] if (lane >> 5) == 0 else 0.0
  return packed


def tile_src_row(trow, off, n, rows_used):
  """rhs_mfma.h::tile_src_row (vectorised over lanes)."""
  inv_n = np.float32(1.0) / np.float32(n)
  sl = ((trow.astype(np.float32) + np.float32(0.5)) * inv_n).astype(np.int32)
  base = sl * n
  q = trow - base + off
  q = np.where(q < 0, q + n, q)
  q = np.where(q >= n, q - n, q)
  return np.where(trow < rows_used, base + q, trow)


def store_tile32(buf, trow, half, acc):
  for qd in range(4):
    for c in range(4):
      buf[trow, 8 * qd + 4 * half + c] = acc[:, 4 * qd + c]


RELU_SHIFT = 64   # dev_params.h: kReluShift


def emulate_tower(un_rows, n, kernels, biases, c_out, relu_shift=RELU_SHIFT):
  """un_rows: [256] already divided by the std.  Returns net [256, 16].

  relu_shift > 0 (the product, dev_params.h kReluShift): activations travel scaled by
  2^-relu_shift -- input-layer weights and every bias row of the tower carry the factor,
  the output layer's weights its inverse (capi.hip: pack_mfma_weights) -- and the relu is
  the VALU's [0, 1] clamp (rhs_mfma.h: activate16); relu_shift = 0: max(x, 0), unscaled."""
  rows_used = (ROWS // n) * n
  dn, up = np.ldexp(1.0, -relu_shift), np.ldexp(1.0, relu_shift)
  relu = (lambda x: np.clip(x, 0.0, 1.0)) if relu_shift else (lambda x: np.maximum(x, 0.0))
  hA = np.zeros((ROWS, HS))
  hB = np.zeros((ROWS, HS))
  w_in = pack_input(kernels[0], biases[0]) * dn
  # ---- input layer -------------------------------------------------------
  for wave in range(4):
    j, half = LANES & 31, LANES >> 5
    for t in range(2):
      trow = wave * 64 + t * 32 + j
      b0 = un_rows[tile_src_row(trow, half - 2, n, rows_used)]
      b1 = un_rows[tile_src_row(trow, half, n, rows_used)]
      b2 = np.where(half == 1, 1.0, un_rows[tile_src_row(trow, 2, n, rows_used)])
      acc = np.zeros((64, 16))
      for s, bop in enumerate((b0, b1, b2)):
        acc = mfma32(w_in[s], bop, acc)
      store_tile32(hA, trow, half, relu(acc))
  src, dst = hA, hB
  # ---- hidden layers -------------------------------------------------------
  for l in range(1, len(kernels) - 1):
    w_h = pack_hidden(kernels[l], biases[l])
    w_h[80] *= dn   # the bias row
    dst[:] = 0
    for wave in range(4):
      j, half = LANES & 31, LANES >> 5
      for t in range(2):
        trow = wave * 64 + t * 32 + j
        acc = np.zeros((64, 16))
        for g in range(20):
          tap, quad = g >> 2, g & 3
          rows = tile_src_row(trow, tap - 2, n, rows_used)
          for c in range(4):
            bop = src[rows, 16 * half + 4 * quad + c]
            acc = mfma32(w_h[4 * g + c], bop, acc)
        acc = mfma32(w_h[80], np.ones(64), acc)
        store_tile32(dst, trow, half, relu(acc))
    src, dst = dst, src
  # ---- output layer (run-time-parameterised kernels: 4 padded groups) ---------
  return emulate_final4(src, n, kernels[-1] * up, biases[-1], c_out, groups=4)


def test_relu_as_scaled_clamp_keeps_the_bits():
  """float32: x -> 2^64 clamp(2^-64 x, 0, 1) IS max(x, 0) for every activation a
  finite trajectory produces (|x| in [2^-62, 2^64]); beyond, it saturates / loses
  subnormal bits -- the documented deviation (dev_params.h: kReluShift)."""
  rs = np.random.RandomState(0)
  x = (rs.randn(200000) * np.exp(rs.uniform(-40, 40, 200000))).astype(np.float32)
  x = x[(np.abs(x) > 2.0 ** -62) & (np.abs(x) < 2.0 ** 63)]
  dn, up = np.float32(2.0 ** -RELU_SHIFT), np.float32(2.0 ** RELU_SHIFT)
  got = np.clip(x * dn, np.float32(0), np.float32(1)) * up
  np.testing.assert_array_equal(got, np.maximum(x, np.float32(0)))
  # weights: scaling by a power of two commutes with float32 rounding of every product
  w = rs.randn(1000).astype(np.float32)
  h = np.abs(rs.randn(1000)).astype(np.float32)
  np.testing.assert_array_equal((w * up) * (h * dn), w * h)


def test_fast_tanh_formula_stays_within_2e7():
  """rhs_mfma.h::fast_tanh: sign(x) (1 - t) / (1 + t), t = exp2(-2 log2(e) |x|), in float32
  with correctly rounded exp2 / reciprocal (the hardware's are within one ulp: <= 1e-7 more);
  below |x| = 1/4, where 1 - t cancels, the odd polynomial
  |x| (1 - x^2/3 + 2 x^4/15 - 17 x^6/315 + 62 x^8/2835).  Absolute error <= 2e-7 everywhere AND relative
  error <= 3e-7 (the quotient alone is 2e-5 off at |x| = 1e-2 and quantised below
  1e-6; tf.tanh keeps relative accuracy)."""
  x = np.concatenate([np.linspace(-12, 12, 400001), np.logspace(-30, 1.2, 200000),
                      -np.logspace(-30, 1.2, 200000), np.linspace(0.24, 0.26, 20001),
                      [0.0, np.inf, -np.inf]]).astype(np.float32)
  f = np.float32
  ax = np.abs(x)

  def fma(a, b, c):   # fmaf: one rounding
    return (a.astype(np.float64) * np.float64(b) + np.float64(c)).astype(np.float32) \
        if np.isscalar(b) else (a.astype(np.float64) * b.astype(np.float64) + np.float64(c)).astype(np.float32)
  with np.errstate(over='ignore', invalid='ignore'):
    t = np.exp2(ax * f(-2.885390081777927)).astype(np.float32)
    q = ((f(1) - t) * (f(1) / (f(1) + t)).astype(np.float32)).astype(np.float32)
    x2 = (x * x).astype(np.float32)
    p4 = fma(x2, f(62.0 / 2835.0), f(-17.0 / 315.0))
    p3 = fma(x2, p4, f(2.0 / 15.0))
    p2 = fma(x2, p3, f(-1.0 / 3.0))
    p1 = fma(x2, p2, f(1.0))
    poly = (ax * p1).astype(np.float32)
  got = np.copysign(np.where(ax < f(0.25), poly, q).astype(np.float32), x)
  want = np.tanh(x.astype(np.float64))
  assert np.abs(got - want).max() < 2e-7
  finite = np.isfinite(x) & (x != 0)
  rel = np.abs(got[finite] - want[finite]) / np.abs(want[finite])
  assert rel.max() < 3e-7, rel.max()
  # the quotient alone loses relative accuracy where the polynomial takes over
  small = finite & (ax < f(0.25))
  assert (np.abs(np.copysign(q, x)[small] - want[small]) / np.abs(want[small])).max() > 1e-5
  assert got[-2] == 1.0 and got[-1] == -1.0 and got[-3] == 0.0


@pytest.mark.parametrize('n,num_layers,c_out', [(64, 3, 9), (32, 3, 11),
                                                 (48, 2, 8), (256, 4, 16),
                                                 (100, 3, 9)])
def test_emulated_tower_matches_oracle(n, num_layers, c_out):
  rs = np.random.RandomState(n + num_layers)
  shapes = [(5, 1 if l == 0 else 32, c_out if l == num_layers - 1 else 32)
            for l in range(num_layers)]
  kernels = [rs.randn(*s).astype(np.float32) * 0.3 for s in shapes]
  biases = [rs.randn(s[2]).astype(np.float32) * 0.1 for s in shapes]
  samples = ROWS // n
  u = rs.randn(samples, n).astype(np.float32)
  spec = dict(standard_deviation=0.8, conv_kernels=kernels, conv_biases=biases,
              num_layers=num_layers, nonlinearity='relu')
  want = oracle.conv_stack(u, spec)                      # [samples, n, c_out]
  un_rows = np.zeros(ROWS)
  un_rows[:samples * n] = (u / np.float32(0.8)).reshape(-1)
  net = emulate_tower(un_rows, n, kernels, biases, c_out)
  got = net[:samples * n, :c_out].reshape(samples, n, c_out)
  np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)
  # padded output channels stay exactly zero
  assert np.all(net[:samples * n, c_out:] == 0)


# ---------------------------------------------------------------------------
# Output layer on v_mfma_f32_4x4x1_16b_f32 with the A block broadcast
# (rhs_mfma.h::final_layer4, capi.hip: "packed4").  Assumed layout (verified
# on hardware by ddd_selftest_mfma_layout): with cbsz = 4 / abid = b, register
# r of lane l accumulates A(lane 4 b + r) * B(lane l).
# ---------------------------------------------------------------------------
FIN4_K = 161


def fin4_regs(groups):
  return (FIN4_K * groups + 15) // 16


def pack_final4(w, b, n_ch, groups=None):   # capi.hip: pack4(groups, renumber)
  groups = groups or (n_ch + 3) // 4
  packed = np.zeros((fin4_regs(4), 64))
  wk = w.reshape(160, -1)
  for k in range(FIN4_K):
    for grp in range(groups):
      q = k * groups + grp
      for r in range(4):
        ch = 4 * grp + r
        if ch >= n_ch:
          continue
        packed[q // 16, 4 * (q % 16) + r] = wk[k, ch] if k < 160 else b[ch]
  return packed, groups


def mfma4(a, bop, acc, abid):
  for r in range(4):
    acc[:, r] += a[4 * abid + r] * bop
  return acc


def emulate_final4(src, n, w, b, n_ch, groups=None):
  """src: [256, 36] hidden activations.  Returns net [256, 4 * groups]."""
  rows_used = (ROWS // n) * n
  packed, groups = pack_final4(w, b, n_ch, groups)
  out = np.zeros((ROWS, 4 * groups))
  for wave in range(4):
    row = wave * 64 + LANES
    acc = [np.zeros((64, 4)) for _ in range(groups)]
    for og in range(40):
      tap, c4 = og // 8, og % 8
      rows = tile_src_row(row, tap - 2, n, rows_used)
      for e in range(4):
        bop = src[rows, 4 * c4 + e]
        for grp in range(groups):
          q = (og * 4 + e) * groups + grp
          acc[grp] = mfma4(packed[q // 16], bop, acc[grp], q % 16)
    for grp in range(groups):
      q = 160 * groups + grp
      acc[grp] = mfma4(packed[q // 16], np.ones(64), acc[grp], q % 16)
    for grp in range(groups):
      out[row, 4 * grp:4 * grp + 4] = acc[grp]
  return out


@pytest.mark.parametrize('n,n_ch', [(64, 12), (64, 14), (32, 11), (256, 11), (128, 12),
                                    (100, 9)])
def test_emulated_final4_matches_direct_conv(n, n_ch):
  rs = np.random.RandomState(n + n_ch)
  w = rs.randn(5, 32, n_ch) * 0.3
  b = rs.randn(n_ch) * 0.1
  samples = ROWS // n
  h = rs.randn(samples, n, 32)
  src = np.zeros((ROWS, HS))
  src[:samples * n, :32] = h.reshape(-1, 32)
  src[:, 32:] = 7.0   # row padding (forcing trig table in the kernel): never read
  got = emulate_final4(src, n, w, b, n_ch)[:samples * n].reshape(samples, n, -1)
  want = np.zeros((samples, n, n_ch))
  for tap in range(5):
    want += np.einsum('bxc,cf->bxf', np.roll(h, 2 - tap, axis=1), w[tap])
  want += b
  np.testing.assert_allclose(got[..., :n_ch], want, rtol=1e-12, atol=1e-12)
  assert np.all(got[..., n_ch:] == 0)


def test_three_instruction_division_equals_ieee_division():
  """rhs_mfma.h scales the input with q = RN(u r), q' = fma(fma(-q, std, u), r, q),
  r = RN(1 / std), instead of the 12-instruction IEEE division sequence
  (model.py:450-451: net = u / std).  Emulated here with exact float64 products
  (a float32 x float32 product is exact in float64, so each fma is one rounding):
  the corrected quotient equals float32 division bit for bit on random inputs,
  the plain reciprocal multiply does not."""
  rs = np.random.RandomState(0)
  for std in (0.7917, 0.594, 0.299, 1.0, 0.123456):   # Burgers / KdV / KS standard deviations
    s = np.float32(std)
    r = np.float32(1.0 / np.float64(s))
    u = np.concatenate([(rs.randn(400000) * 2).astype(np.float32),
                        np.float32([0.0, -0.0, 1e-30, -1e30, 1e37])])
    q = (u * r).astype(np.float32)
    e = (np.float64(u) - np.float64(q) * np.float64(s)).astype(np.float32)
    q2 = (np.float64(q) + np.float64(e) * np.float64(r)).astype(np.float32)
    want = (u / s).astype(np.float32)
    np.testing.assert_array_equal(q2, want)
    # a quotient that overflows (|u| > FLT_MAX std: a diverged state) comes out
    # non-finite either way -- Inf from the division, NaN from the correction
    # step (Inf - Inf); divergence is reported as non-finite rows, not clamped
    with np.errstate(all='ignore'):
      big = np.float32(3.4e38)
      qb = np.float32(big * r)
      eb = np.float32(np.float64(big) - np.float64(qb) * np.float64(s))
      assert not np.isfinite(np.float32(np.float64(qb) + np.float64(eb) * np.float64(r))) \
          or np.isfinite(big / s)
    if std != 1.0:
      assert np.mean(q != want) > 0.05     # the uncorrected product is NOT the quotient


# ---------------------------------------------------------------------------
# Towers with streamed weights (rhs_mfma.h Tower<K, CB>: input_layer_big,
# hidden_layer_stream, final_layer4<NG, TW>; capi.hip pack_mfma_weights `big` branch
# + embed_tower): K taps, CB blocks of 32 channels, activations in rows of 32 CB + 4.
# ---------------------------------------------------------------------------
def pack_input_big(w, b, taps, blocks):          # [h][s][lane]
  steps = (taps + 2) // 2
  packed = np.zeros((blocks, steps, 64))
  for h in range(blocks):
    for s in range(steps):
      for lane in range(64):
        k, ch = 2 * s + (lane >> 5), 32 * h + (lane & 31)
        packed[h, s, lane] = w[k, 0, ch] if k < taps else (b[ch] if k == taps else 0.0)
  return packed


def pack_hidden_stream(w, b, taps, blocks):      # [group][block][lane][4] + bias [block][lane]
  chans = 32 * blocks
  groups = taps * chans // 8
  packed = np.zeros((groups, blocks, 64, 4))
  for g in range(groups):
    for h in range(blocks):
      for lane in range(64):
        for e in range(4):
          s = 4 * g + e
          tap, cb, jj = s // (16 * blocks), (s // 16) % blocks, s % 16
          cin, cout = 32 * cb + 16 * (lane >> 5) + jj, 32 * h + (lane & 31)
          packed[g, h, lane, e] = w[tap, cin, cout]
  bias = np.zeros((blocks, 64))
  for h in range(blocks):
    bias[h, :32] = b[32 * h:32 * h + 32]
  return packed, bias


def embed(kernels, biases, taps, chans):         # capi.hip: embed_tower
  out_k, out_b = [], []
  for l, (w, b) in enumerate(zip(kernels, biases)):
    k_true = w.shape[0]
    shift = (taps - 1) // 2 - k_true // 2
    cin = 1 if l == 0 else chans
    cout = w.shape[2] if l == len(kernels) - 1 else chans
    wp = np.zeros((taps, cin, cout), w.dtype)
    wp[shift:shift + k_true, :w.shape[1], :w.shape[2]] = w
    bp = np.zeros(cout, b.dtype)
    bp[:b.shape[0]] = b
    out_k.append(wp)
    out_b.append(bp)
  return out_k, out_b


def emulate_big_tower(un_rows, n, kernels, biases, taps, blocks, groups):
  """One-wave geometry generalised to 256 rows (four wavefronts): the operand
  gathers of input_layer_big / hidden_layer_stream / final_layer4<NG, TW>."""
  chans, hs, left = 32 * blocks, 32 * blocks + 4, taps // 2
  rows_used = (ROWS // n) * n
  # (relu = the [0, 1] clamp on activations scaled by 2^-RELU_SHIFT, as in emulate_tower)
  dn, up = np.ldexp(1.0, -RELU_SHIFT), np.ldexp(1.0, RELU_SHIFT)
  relu = lambda x: np.clip(x, 0.0, 1.0)
  bufs = [np.zeros((ROWS, hs)), np.zeros((ROWS, hs))]
  j, half = LANES & 31, LANES >> 5
  w_in = pack_input_big(kernels[0], biases[0], taps, blocks) * dn
  steps = (taps + 2) // 2
  for wave in range(4):
    for t in range(2):
      trow = wave * 64 + t * 32 + j
      ops = []
      for s in range(steps):
        if 2 * s < taps - 1:
          k = np.where(half == 1, 2 * s + 1, 2 * s)
          ops.append(un_rows[tile_src_row(trow, k - left, n, rows_used)])
        else:   # last step: tap K - 1 beside the bias row
          ops.append(np.where(half == 1, 1.0,
                              un_rows[tile_src_row(trow, taps - 1 - left, n, rows_used)]))
      for h in range(blocks):
        acc = np.zeros((64, 16))
        for s in range(steps):
          acc = mfma32(w_in[h, s], ops[s], acc)
        for qd in range(4):
          for c in range(4):
            bufs[0][trow, 32 * h + 8 * qd + 4 * half + c] = relu(acc)[:, 4 * qd + c]
  src, dst = bufs
  for l in range(1, len(kernels) - 1):
    packed, bias = pack_hidden_stream(kernels[l], biases[l], taps, blocks)
    bias = bias * dn
    dst[:] = 0
    for wave in range(4):
      for t in range(2):
        trow = wave * 64 + t * 32 + j
        acc = [np.zeros((64, 16)) for _ in range(blocks)]
        for g in range(packed.shape[0]):
          tap, cb, quad = g // (4 * blocks), (g // 4) % blocks, g % 4
          rows = tile_src_row(trow, tap - left, n, rows_used)
          for e in range(4):
            bop = src[rows, 32 * cb + 16 * half + 4 * quad + e]
            for h in range(blocks):
              acc[h] = mfma32(packed[g, h, :, e], bop, acc[h])
        for h in range(blocks):
          acc[h] = mfma32(bias[h], np.ones(64), acc[h])
          for qd in range(4):
            for c in range(4):
              dst[trow, 32 * h + 8 * qd + 4 * half + c] = relu(acc[h])[:, 4 * qd + c]
    src, dst = dst, src
  # output layer: k = tap * chans + c in natural order, K C + 1 reduction steps
  w, b = kernels[-1], biases[-1]
  n_ch, kc = w.shape[2], taps * chans
  fin_k = kc + 1
  packed = np.zeros(((fin_k * groups + 15) // 16, 64))
  wk = w.reshape(kc, n_ch)
  for k in range(fin_k):
    for grp in range(groups):
      q = k * groups + grp
      for r in range(4):
        ch = 4 * grp + r
        if ch < n_ch:
          packed[q // 16, 4 * (q % 16) + r] = up * wk[k, ch] if k < kc else b[ch]
  out = np.zeros((ROWS, 4 * groups))
  per_tap = chans // 4
  for wave in range(4):
    row = wave * 64 + LANES
    acc = [np.zeros((64, 4)) for _ in range(groups)]
    for og in range(taps * per_tap):
      tap, c4 = og // per_tap, og % per_tap
      rows = tile_src_row(row, tap - left, n, rows_used)
      for e in range(4):
        bop = src[rows, 4 * c4 + e]
        for grp in range(groups):
          q = (og * 4 + e) * groups + grp
          acc[grp] = mfma4(packed[q // 16], bop, acc[grp], q % 16)
    for grp in range(groups):
      q = kc * groups + grp
      acc[grp] = mfma4(packed[q // 16], np.ones(64), acc[grp], q % 16)
      out[row, 4 * grp:4 * grp + 4] = acc[grp]
  return out


@pytest.mark.parametrize('n,num_layers,c_out,k_true,f_true,taps,blocks', [
    (64, 3, 9, 7, 32, 7, 1), (64, 3, 12, 5, 64, 5, 2), (32, 3, 8, 3, 32, 3, 1),
    (96, 4, 11, 6, 20, 7, 1),      # embedded: 6 taps x 20 filters in 7 x 32
    (128, 2, 9, 4, 40, 5, 2),      # embedded: 4 taps x 40 filters in 5 x 64, no hidden layer
    (64, 3, 10, 7, 64, 7, 2),      # 7 taps x 64 filters (the kernel rolls its hidden layers over the taps;
    (64, 3, 8, 6, 48, 7, 2),       #   same packed layout), and 6 x 48 embedded in it
])
def test_emulated_streamed_towers_match_oracle(n, num_layers, c_out, k_true, f_true, taps, blocks):
  rs = np.random.RandomState(n + taps + blocks)
  shapes = [(k_true, 1 if l == 0 else f_true, c_out if l == num_layers - 1 else f_true)
            for l in range(num_layers)]
  kernels = [rs.randn(*s).astype(np.float32) * 0.3 for s in shapes]
  biases = [rs.randn(s[2]).astype(np.float32) * 0.1 for s in shapes]
  samples = ROWS // n
  u = rs.randn(samples, n).astype(np.float32)
  spec = dict(standard_deviation=0.8, conv_kernels=kernels, conv_biases=biases,
              num_layers=num_layers, nonlinearity='relu')
  want = oracle.conv_stack(u, spec)                      # the TRUE net
  un_rows = np.zeros(ROWS)
  un_rows[:samples * n] = (u / np.float32(0.8)).reshape(-1)
  pk, pb = embed(kernels, biases, taps, 32 * blocks)
  groups = (c_out + 3) // 4
  net = emulate_big_tower(un_rows, n, pk, pb, taps, blocks, groups)
  got = net[:samples * n, :c_out].reshape(samples, n, c_out)
  np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-5)
  assert np.all(net[:samples * n, c_out:] == 0)


def wide_slot(g):
  """rhs_mfma.h: wide_slot -- channels per derivative of the wide flavour's folded output layer."""
  return 8 if g <= 8 else g


def fold_wide_output_layer(w_nat, b_nat, nullspaces, acc_biases, g):
  """capi.hip::pack_mfma_weights, wide flavour (always folded): output channel
  wide_slot(G) d + g' of the folded layer = sum_j W[:, start_d + j] * nullspace_d[j, g']
  (float64 accumulation, rounded once), bias row = accuracy bias + projected conv bias."""
  kc, slot = w_nat.shape[0], wide_slot(g)
  cols = 36
  wf, bf = np.zeros((kc, cols), np.float32), np.zeros(cols, np.float32)
  start = 0
  for d, (ns, ab) in enumerate(zip(nullspaces, acc_biases)):
    if ns is None:   # polynomial_accuracy_order 0: channel G d + g' IS the coefficient
      wf[:, slot * d:slot * d + g] = w_nat[:, g * d:g * d + g]
      bf[slot * d:slot * d + g] = b_nat[g * d:g * d + g]
      continue
    stop = start + ns.shape[0]
    wf[:, slot * d:slot * d + g] = (w_nat[:, start:stop].astype(np.float64) @ ns.astype(np.float64)).astype(np.float32)
    bf[slot * d:slot * d + g] = (ab.astype(np.float64) + b_nat[start:stop].astype(np.float64) @ ns.astype(np.float64)).astype(np.float32)
    start = stop
  return wf, bf


@pytest.mark.parametrize('g,free,direct', [(9, (7, 6, 4), False), (12, (10, 9), False),
                                          (10, (8, 7, 5), False), (7, None, True), (9, None, True)])
def test_wide_fold_equals_projection(g, free, direct):
  """The wide kernels have no projection code: their output layer is folded on the host.
  The folded layer applied to the last hidden activations must give the coefficients
  bias + net @ nullspace (polynomials.py:275-277) -- or the net's own channels
  (model.py:460-475) -- in slots of wide_slot(G) per derivative, zeros elsewhere."""
  rs = np.random.RandomState(g)
  derivs = 3 if direct or len(free) == 3 else 2
  c_out = derivs * g if direct else sum(free)
  kc = 160
  w_nat = (rs.randn(kc, c_out) * 0.2).astype(np.float32)
  b_nat = (rs.randn(c_out) * 0.1).astype(np.float32)
  hidden = np.maximum(rs.randn(50, kc), 0).astype(np.float32)     # relu activations, [rows][tap, cin]
  if direct:
    nullspaces, acc_biases = [None] * derivs, [None] * derivs
  else:
    nullspaces = [rs.randn(f, g).astype(np.float32) for f in free]
    acc_biases = [rs.randn(g).astype(np.float32) for _ in free]
  wf, bf = fold_wide_output_layer(w_nat, b_nat, nullspaces, acc_biases, g)
  got = hidden.astype(np.float64) @ wf.astype(np.float64) + bf
  net = hidden.astype(np.float64) @ w_nat.astype(np.float64) + b_nat
  slot, start = wide_slot(g), 0
  used = np.zeros(36, bool)
  for d in range(derivs):
    if direct:
      want = net[:, g * d:g * d + g]
    else:
      stop = start + free[d]
      want = acc_biases[d] + net[:, start:stop] @ nullspaces[d].astype(np.float64)
      start = stop
    np.testing.assert_allclose(got[:, slot * d:slot * d + g], want, rtol=2e-5, atol=2e-5)
    used[slot * d:slot * d + g] = True
  assert derivs * slot - (slot - g) <= 36            # the channels the kernel carries
  assert np.all(wf[:, ~used] == 0) and np.all(bf[~used] == 0)


# ---------------------------------------------------------------------------
# Four 16-row wavefronts per 64-row group (rhs_mfma.h kQuad): every layer on
# v_mfma_f32_16x16x4_f32 -- lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15];
# register r of lane l holds D[4 (l >> 4) + r][l & 15] (cdna_hip_programming.md section 3).
# Transcribes capi.hip's quad packing, rhs_mfma.h's lane_offsets (kWR == 16) and
# input_layer_quad / hidden_layer_quad / final_layer_quad.
# ---------------------------------------------------------------------------
def mfma16(a, b, acc):
  A = a.reshape(4, 16).T.astype(np.float64)      # [i, k]
  B = b.reshape(4, 16).astype(np.float64)        # [k, j]
  D = A @ B
  for r in range(4):
    acc[:, r] += D[4 * (LANES >> 4) + r, LANES & 15]
  return acc


def quad_channel_float(c):     # rhs_mfma.h: position of channel c in an LDS row
  return 4 * (4 * (c >> 4) + (c & 3)) + ((c & 15) >> 2)


def pack_quad(kernels, biases, w_out, b_out, n_ch, dn, up):
  """capi.hip: the w_quad rows ([2][2] input, [2][41] hidden, [41] output) x 64 lanes."""
  w0, b0, w1, b1 = kernels[0], biases[0], kernels[1], biases[1]
  rows = np.zeros((4 + 82 + 41, 64))
  for chh in range(2):
    for lane in range(64):
      sg, cout = lane >> 4, 16 * chh + (lane & 15)
      rows[chh * 2 + 0, lane] = dn * w0[sg, 0, cout]
      rows[chh * 2 + 1, lane] = dn * w0[4, 0, cout] if sg == 0 else dn * b0[cout] if sg == 1 else 0.0
      for s2 in range(40):
        tap, i = s2 // 8, s2 % 8
        cin = (sg >> 1) + 16 * (sg & 1) + 2 * i
        rows[4 + chh * 41 + s2, lane] = w1[tap, cin, cout]
      rows[4 + chh * 41 + 40, lane] = dn * b1[cout] if sg == 0 else 0.0
  w_flat = w_out.reshape(160, -1)
  for s2 in range(41):
    for lane in range(64):
      k, ch = 4 * s2 + (lane >> 4), lane & 15
      if ch >= n_ch or k > 160:
        continue
      rows[4 + 82 + s2, lane] = up * w_flat[k, ch] if k < 160 else b_out[ch]
  return rows


def emulate_tower_quad(un64, n, kernels, biases, n_ch, relu_shift=RELU_SHIFT):
  """One 64-row group on four wavefronts.  un64: [64] = u / std.  Returns net [64, 16]."""
  dn, up = np.ldexp(1.0, -relu_shift), np.ldexp(1.0, relu_shift)
  relu = (lambda x: np.clip(x, 0.0, 1.0)) if relu_shift else (lambda x: np.maximum(x, 0.0))
  wq = pack_quad(kernels, biases, kernels[2], biases[2], n_ch, dn, up)
  hA, hB = np.full((64, HS), np.nan), np.full((64, HS), np.nan)
  sg, j16 = LANES >> 4, LANES & 15

  def tap_row(trow, off):      # rows of (pos + off) mod n inside the row's sample (n | 64)
    base = trow & ~(n - 1)
    return ((trow + off) & (n - 1)) | base

  def store16(buf, trow, chh, acc):   # channels 16 chh + 4 sg + r -> block (chh, r), element sg
    for r in range(4):
      buf[trow, 16 * chh + 4 * r + sg] = acc[:, r]
      assert (quad_channel_float(16 * chh + 4 * sg + r) == 16 * chh + 4 * r + sg).all()

  for wave in range(4):        # input layer
    ph, chh = wave & 1, wave >> 1
    for t in range(2):
      trow = 32 * ph + 16 * t + j16
      b0 = un64[tap_row(trow, sg - 2)]
      b1 = np.where(LANES < 16, un64[tap_row(trow, 2)], 1.0)
      acc = np.zeros((64, 4))
      acc = mfma16(wq[chh * 2 + 0], b0, acc)
      acc = mfma16(wq[chh * 2 + 1], b1, acc)
      store16(hA, trow, chh, relu(acc))
  for wave in range(4):        # hidden layer
    ph, chh = wave & 1, wave >> 1
    for t in range(2):
      trow = 32 * ph + 16 * t + j16
      acc = np.zeros((64, 4))
      blk = 4 * (sg & 1) + (sg >> 1)
      for tap in range(5):
        rows = tap_row(trow, tap - 2)
        qa = [hA[rows, 4 * blk + e] for e in range(4)]
        qb = [hA[rows, 4 * (blk + 2) + e] for e in range(4)]
        for i in range(8):
          bop = (qa if i % 2 == 0 else qb)[i // 2]
          acc = mfma16(wq[4 + chh * 41 + 8 * tap + i], bop, acc)
      acc = mfma16(wq[4 + chh * 41 + 40], np.ones(64), acc)
      store16(hB, trow, chh, relu(acc))
  net = np.zeros((64, 16))
  for wave in range(4):        # output layer: the wavefront's own 16 rows
    row = 16 * wave + j16
    acc = np.zeros((64, 4))
    for tap in range(5):
      rows = tap_row(row, tap - 2)
      q0 = [hB[rows, 4 * sg + e] for e in range(4)]
      q1 = [hB[rows, 4 * (4 + sg) + e] for e in range(4)]
      for i in range(8):
        acc = mfma16(wq[4 + 82 + 8 * tap + i], (q0 + q1)[i], acc)
    acc = mfma16(wq[4 + 82 + 40], np.ones(64), acc)
    for r in range(4):
      net[row, 4 * sg + r] = acc[:, r]     # lane (sg, row) holds channels 4 sg .. 4 sg + 3
  return net


@pytest.mark.parametrize('n,c_out', [(64, 12), (32, 9), (16, 14), (8, 11)])
def test_quad_flavour_data_movement(n, c_out):
  """The four-wavefront flavour computes the tower the one-wavefront emulation (and the
  oracle's conv stack) computes."""
  rs = np.random.RandomState(3 * n + c_out)
  shapes = [(5, 1, 32), (5, 32, 32), (5, 32, c_out)]
  kernels = [rs.randn(*s).astype(np.float32) * 0.3 for s in shapes]
  biases = [rs.randn(s[2]).astype(np.float32) * 0.1 for s in shapes]
  samples = 64 // n
  u = rs.randn(samples, n).astype(np.float32)
  un = (u / np.float32(0.8)).reshape(-1).astype(np.float64)
  got = emulate_tower_quad(un, n, kernels, biases, c_out)
  spec = dict(standard_deviation=0.8, conv_kernels=kernels, conv_biases=biases,
              num_layers=3, nonlinearity='relu')
  want = oracle.conv_stack(u, spec).reshape(64, c_out)
  np.testing.assert_allclose(got[:, :c_out], want, rtol=2e-4, atol=2e-5)
  assert np.all(got[:, c_out:] == 0)
  # against the one-wavefront emulation (float64 both, different association only)
  un_rows = np.zeros(ROWS)
  un_rows[:64] = un
  one = emulate_tower(un_rows, n, kernels, biases, c_out)[:64]
  np.testing.assert_allclose(got[:, :c_out], one[:, :c_out], rtol=1e-9, atol=1e-12)


def test_quad_flavour_reduction_order_is_the_one_wavefront_kernels():
  """Bit-identity rests on the ORDER of every fma chain (an f32 MFMA is an fmaf chain over
  its reduction slots in slot order): per output element the sequence of (tap, channel)
  terms must be the one-wavefront kernel's."""
  # hidden layer, one-wavefront kernel: step s = 16 tap + jj, slots (half 0, half 1) = cin jj, 16 + jj
  one = [(s // 16, 16 * half + s % 16) for s in range(80) for half in (0, 1)]
  # quad: step 8 tap + i, slots sg = 0..3: cin = (sg >> 1) + 16 (sg & 1) + 2 i
  quad = [(s // 8, (sg >> 1) + 16 * (sg & 1) + 2 * (s % 8)) for s in range(40) for sg in range(4)]
  assert one == quad
  # output layer: final_layer4 issues k = 32 tap + c one per instruction in natural order
  assert [4 * s + sg for s in range(40) for sg in range(4)] == list(range(160))
  # input layer: (tap 0, tap 1), (tap 2, tap 3), (tap 4, bias)  ==  (taps 0..3), (tap 4, bias, 0, 0)
  assert [2 * s + h for s in range(3) for h in (0, 1)] == [0, 1, 2, 3, 4, 5]


# ---------------------------------------------------------------------------
# Nets of up to 16 filters on 16-channel tiles (rhs_mfma.h Tile16Tower, round 6): ONE wavefront
# per 64-row group, every layer of the tower on v_mfma_f32_16x16x4_f32, four position tiles.
# Transcribes capi.hip's d_w_t16 / d_w_final4_half packing, lane_offsets<..., kTile16>,
# input_layer_t16 / hidden_layer_t16 / final_layer4_t16.
# ---------------------------------------------------------------------------
def t16_channel_float(c):      # position of channel c (< 16) in an LDS row
  return 4 * (c & 3) + (c >> 2)


def pack_t16(kernels, biases, dn):
  """capi.hip: [2] input + [21] hidden rows x 64 lanes from the net EMBEDDED in 32 filters."""
  w0, b0, w1, b1 = kernels[0], biases[0], kernels[1], biases[1]
  rows = np.zeros((2 + 21, 64))
  for lane in range(64):
    sg, cout = lane >> 4, lane & 15
    rows[0, lane] = dn * w0[sg, 0, cout]
    rows[1, lane] = dn * w0[4, 0, cout] if sg == 0 else dn * b0[cout] if sg == 1 else 0.0
    for s2 in range(20):
      tap, e = s2 // 4, s2 % 4
      rows[2 + s2, lane] = w1[tap, 4 * e + sg, cout]
    rows[2 + 20, lane] = dn * b1[cout] if sg == 0 else 0.0
  return rows


def emulate_tower_t16(un64, n, kernels, biases, n_ch, relu_shift=RELU_SHIFT):
  """kernels / biases: the net embedded in 5 taps x 32 filters (channels >= 16 zero)."""
  dn, up = np.ldexp(1.0, -relu_shift), np.ldexp(1.0, relu_shift)
  relu = (lambda x: np.clip(x, 0.0, 1.0)) if relu_shift else (lambda x: np.maximum(x, 0.0))
  wq = pack_t16(kernels, biases, dn)
  hA, hB = np.full((64, HS), np.nan), np.full((64, HS), np.nan)
  sg, j16 = LANES >> 4, LANES & 15

  def tap_row(trow, off):
    base = trow & ~(n - 1)
    return ((trow + off) & (n - 1)) | base

  def store16(buf, trow, acc):   # lane (j, sg) holds channels 4 sg + r -> float 4 r + sg (store_tile16 at + 4 sg)
    for r in range(4):
      buf[trow, 4 * r + sg] = acc[:, r]
      assert (t16_channel_float(4 * sg + r) == 4 * r + sg).all()

  for t in range(4):             # input layer: taps 0..3 | tap 4, bias, 0, 0
    trow = 16 * t + j16
    b0 = un64[tap_row(trow, sg - 2)]
    b1 = np.where(LANES < 16, un64[tap_row(trow, 2)], 1.0)
    acc = np.zeros((64, 4))
    acc = mfma16(wq[0], b0, acc)
    acc = mfma16(wq[1], b1, acc)
    store16(hA, trow, relu(acc))
  for t in range(4):             # hidden layer: step 4 tap + e, slot sg = channel 4 e + sg
    trow = 16 * t + j16
    acc = np.zeros((64, 4))
    for tap in range(5):
      rows = tap_row(trow, tap - 2)
      for e in range(4):         # ONE ds_read_b128 at float 4 sg: elements e = 0..3
        bop = hA[rows, 4 * sg + e]
        assert (4 * sg + e == t16_channel_float(4 * e + sg)).all()
        acc = mfma16(wq[2 + 4 * tap + e], bop, acc)
    acc = mfma16(wq[2 + 20], np.ones(64), acc)
    store16(hB, trow, relu(acc))
  # output layer on the 4x4x1 MFMAs, lane == row: four float4 blocks per tap row, channels
  # picked in natural order: channel c = element (c >> 2) of block (c & 3)
  w_flat = kernels[2].reshape(5, 32, -1)
  net = np.zeros((64, 16))
  rows64 = np.arange(64)
  for ch in range(n_ch):
    acc = np.zeros(64)
    for tap in range(5):
      src = tap_row(rows64, tap - 2)
      for c in range(16):
        acc = acc + up * w_flat[tap, c, ch] * hB[src, 4 * (c & 3) + (c >> 2)]
    net[:, ch] = acc + biases[2][ch]
  return net


@pytest.mark.parametrize('n,filters,c_out', [(64, 16, 12), (32, 12, 9), (16, 5, 14), (8, 16, 11)])
def test_tile16_data_movement(n, filters, c_out):
  """The 16-channel tiles compute the tower the oracle's conv stack computes for the TRUE net
  (and the one-wavefront emulation for its embedding in 32 filters)."""
  rs = np.random.RandomState(7 * n + filters + c_out)
  true_shapes = [(5, 1, filters), (5, filters, filters), (5, filters, c_out)]
  kernels = [rs.randn(*s).astype(np.float32) * 0.3 for s in true_shapes]
  biases = [rs.randn(s[2]).astype(np.float32) * 0.1 for s in true_shapes]
  emb_k = [np.zeros((5, 1, 32), np.float32), np.zeros((5, 32, 32), np.float32),
           np.zeros((5, 32, c_out), np.float32)]
  emb_b = [np.zeros(32, np.float32), np.zeros(32, np.float32), biases[2]]
  emb_k[0][:, :, :filters] = kernels[0]
  emb_k[1][:, :filters, :filters] = kernels[1]
  emb_k[2][:, :filters, :] = kernels[2]
  emb_b[0][:filters] = biases[0]
  emb_b[1][:filters] = biases[1]
  samples = 64 // n
  u = rs.randn(samples, n).astype(np.float32)
  un = (u / np.float32(0.8)).reshape(-1).astype(np.float64)
  got = emulate_tower_t16(un, n, emb_k, emb_b, c_out)
  spec = dict(standard_deviation=0.8, conv_kernels=kernels, conv_biases=biases,
              num_layers=3, nonlinearity='relu')
  want = oracle.conv_stack(u, spec).reshape(64, c_out)
  np.testing.assert_allclose(got[:, :c_out], want, rtol=2e-4, atol=2e-5)
  un_rows = np.zeros(ROWS)
  un_rows[:64] = un
  one = emulate_tower(un_rows, n, emb_k, emb_b, c_out)[:64]
  np.testing.assert_allclose(got[:, :c_out], one[:, :c_out], rtol=1e-9, atol=1e-12)


def test_tile16_reduction_order_is_the_embedded_evaluations():
  """Bit-identity with the zero-padded embedding rests on the order of every fma chain."""
  # hidden layer, embedded one-wavefront kernel: step 16 tap + jj, slots cin = jj, 16 + jj; the
  # channels >= 16 carry zero weights and zero activations: what remains is c = 0..15 per tap
  embedded = [(tap, c) for tap in range(5) for c in range(16)]
  # tiles: step 4 tap + e, slots sg = 0..3: channel 4 e + sg
  tiles = [(s // 4, 4 * (s % 4) + sg) for s in range(20) for sg in range(4)]
  assert tiles == embedded
  # output layer: k = 16 tap + c in natural order, operand = element c >> 2 of block c & 3
  assert sorted(4 * (c & 3) + (c >> 2) for c in range(16)) == list(range(16))
