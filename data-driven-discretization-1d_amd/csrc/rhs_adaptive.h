// Batched adaptive Bogacki-Shampine integration on the device: SciPy's RK23
// (the reference's production integrator, integrate.py:143-169:
// solve_ivp(..., t_eval=times, max_step=0.01, method='RK23')) with ONE step-size
// controller per sample, inside the persistent MFMA kernel.
//
// What is restated (scipy/integrate/_ivp, third-party, unpinned by the
// reference; the per-sample scalar part lives in rk23.h;
// oracle/oracle.py::rk23_adaptive is the same restatement in NumPy, pinned
// against the installed SciPy):
//   RungeKutta.__init__      f0 = fun(t0, y0); select_initial_step (order 2)
//   RungeKutta._step_impl    min_step, clamp to [min_step, max_step], attempt loop,
//                            error norm = RMS(err / (atol + rtol max(|y|, |y_new|))),
//                            factor = SAFETY norm^(-1/3) in [0.2, 10], no growth
//                            after a rejection, TOO_SMALL_STEP failure
//   rk_step                  FSAL Bogacki-Shampine stages, K in float64
//   solve_ivp + RkDenseOutput  cubic dense output at every t_eval <= t
// The state, the stage combinations, the controller and the dense output are
// float64 (SciPy holds y in float64); the right-hand side is the float32
// evaluation of rhs_mfma.h, fed float32(y) and float32(t) exactly like the
// reference's TF placeholders (integrate.py:57-60).
//
// Shape: the work decomposition of integrate_kernel (rhs_mfma.h): a workgroup
// owns whole samples, lane == grid point.  Every sample carries its own t, h,
// status and evaluation count; the samples of one workgroup share the
// evaluations (all are at the same stage of *some* attempt), finished samples
// idle until the last sample of their workgroup is done.  One call site of
// eval_rhs (a phase counter drives what its input and result mean): the
// evaluation is ~3 k instructions and must not be replicated five times.
#pragma once
#include "rhs_mfma.h"
#include "rk23.h"
#ifdef DDD_PROBES   // libddd1d_probe.so: phase stamps (profiles/tools/adaptive_phase_trace.py)
#include "rhs_adaptive_trace.h"
#else
#define DDD_ADAPT_TRACE_SETUP do {} while (0)
#define DDD_ADAPT_STAMP(K) do {} while (0)
#endif

namespace ddd {
namespace mfma {

// Sum of `v` over the lanes of this lane's sample, identical (bitwise) on all of them.
template <int kRows, int kWR>
__device__ __forceinline__ double sample_sum(const DevParams& p, const Lane& ln, double v,
                                             double* red) {
  if (kRows == kWR) {
    // one wavefront, N | 64 (a power of two): xor butterfly inside aligned
    // groups of N lanes; a + b == b + a, so every lane ends with the same bits.
    // ds_bpermute at (lane ^ m) * 4 directly: __shfl_xor spends nine VALU instructions per
    // step on its width check and index math (54 per error norm, on lanes the MFMA pipe
    // shares); here one v_xor per step -- lane ^ m never leaves an aligned group of N > m.
#if !DDD_ADAPTIVE_BUTTERFLY
    for (int m = 1; m < p.N; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
#endif
    const int lane4 = ln.lane << 2;
    for (int m = 1; m < p.N; m <<= 1) {
      const int src = lane4 ^ (m << 2);
      const long long bits = __double_as_longlong(v);
      const int lo = __builtin_amdgcn_ds_bpermute(src, (int)bits);
      const int hi = __builtin_amdgcn_ds_bpermute(src, (int)(bits >> 32));
      v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
  }
  // four-wave groups: `red` aliases Shared::un + Shared::flux (free between two
  // evaluations) -- slower wavefronts may still be reading the flux exchange
  __syncthreads();
  if constexpr (kWR == 16) {
    // four 16-row wavefronts per 64-row group (kQuad; N | 64, every 16-lane quarter of a
    // wavefront carries the same 16 rows): xor butterfly over the sample's rows inside the
    // quarter, then the partial sums of the sample's wavefronts combined as the
    // one-wavefront kernel's butterfly combines them -- (w0 + w1) + (w2 + w3) -- so the
    // controller sees the same bits whatever the geometry
    const int lane4 = ln.lane << 2;
    const int inner = p.N < 16 ? p.N : 16;
    for (int m = 1; m < inner; m <<= 1) {
      const int src = lane4 ^ (m << 2);
      const long long bits = __double_as_longlong(v);
      const int lo = __builtin_amdgcn_ds_bpermute(src, (int)bits);
      const int hi = __builtin_amdgcn_ds_bpermute(src, (int)(bits >> 32));
      v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    if (p.N <= 16) return v;   // (wave-uniform: whole samples inside the quarter)
    if (ln.lane == 0) red[ln.wave] = v;
    __syncthreads();
    double s;
    if (p.N == 32) s = red[ln.wave & ~1] + red[ln.wave | 1];
    else s = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return s;
  } else
  if ((p.N & 63) == 0) {
    // a sample is N / 64 whole wavefronts (KS N = 256: all four): xor butterfly
    // inside each wavefront, then the sample's wave totals in a fixed order --
    // 6 shuffles + <= 4 adds instead of a serial N-term loop per lane
    // (rk23.h: block_sum256 is the N = 256 case).  (wave-uniform branch)
    for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
    if (ln.lane == 0) red[ln.wave] = v;
    __syncthreads();
    const int w0 = ln.base >> 6, nw = p.N >> 6;
    double s = red[w0];
    for (int i = 1; i < nw; ++i) s += red[w0 + i];
    __syncthreads();
    return s;
  } else {
    red[ln.row] = v;
    __syncthreads();
    double s = 0.0;
    const double* src = red + ln.base;   // spare rows: base 0, result unused
    for (int i = 0; i < p.N; ++i) s += src[i];
    __syncthreads();
    return s;
  }
}

// The per-sample controllers as they live in LDS (40 bytes: the 256-row kernels
// have 1.6 KB of LDS to spare next to Shared<256> if two workgroups are to share a CU).
struct PackedControl {
  double t, t_new, h, h_abs;
  int nfev;
  int bits;   // (status + 2) | rejected << 2 | ti << 3
  __device__ __forceinline__ void store(const rk23::Control& c) {
    t = c.t; t_new = c.t_new; h = c.h; h_abs = c.h_abs; nfev = c.nfev;
    bits = (c.status + 2) | ((c.rejected ? 1 : 0) << 2) | (c.ti << 3);
  }
  __device__ __forceinline__ rk23::Control load() const {
    rk23::Control c;
    c.t = t; c.t_new = t_new; c.h = h; c.h_abs = h_abs; c.nfev = nfev;
    const int b = bits;
    c.status = (b & 3) - 2; c.rejected = ((b >> 2) & 1) != 0; c.ti = b >> 3;
    return c;
  }
  __device__ __forceinline__ int status() const { return (bits & 3) - 2; }
};
static_assert(rk23::ATTEMPT_LIMIT == -2 && rk23::RUNNING == 1, "status + 2 fits two bits");

template <int kRows>
struct AdaptiveShared {
  PackedControl ctl[kRows / 8];      // one per sample of the group (N >= 8)
  int attempts_of[kRows / 8];        // saturating (2^31 attempts of one sample = hours)
  int vote[2];
};
// (budgets: 8 x 64-row workgroups per CU leave 376 bytes next to Shared<64>, 2 x 256-row
// workgroups 1600 next to Shared<256>)
static_assert(sizeof(Shared<64>) + sizeof(AdaptiveShared<64>) <= 20 * 1024, "8 groups per CU");

#ifndef DDD_ADAPTIVE_LEAN
#define DDD_ADAPTIVE_LEAN 0   // A/B (profiles/r4_ablation.txt): bit 0 = 64-row groups lean, bit 1 = 256-row
#endif
// eval_rhs's kLean for the adaptive kernels: 1 = output-layer weights and cos / sin table
// fetched per evaluation (the A/B above), 2 = the table only (two ds_read_b128 from the
// LDS row padding instead of 12 resident registers: the controller's live values -- state
// and three stage derivatives in float64 / float32, the next output time -- need them)
#ifndef DDD_ADAPTIVE_TRIG_LDS
#define DDD_ADAPTIVE_TRIG_LDS 0   // measured (gpurun_out/r5d): resident 77.7 / 72.1 / 66.4 %, LDS / L2 77.2 / 71.8 / 66.3 %
#endif
// A/B switches of round 5's controller slimming (profiles/r5_ablation.txt)
#ifndef DDD_ADAPTIVE_SHORTCUT
#define DDD_ADAPTIVE_SHORTCUT 1   // stages 2 / 3 without controller load / store / vote
#endif
#ifndef DDD_ADAPTIVE_PREFETCH
#define DDD_ADAPTIVE_PREFETCH 1   // next output time requested during stage 3
#endif
#ifndef DDD_ADAPTIVE_SPECULATE
#define DDD_ADAPTIVE_SPECULATE 0   // first-stage forcing sums prepared across the error test (round 6):
                                   // measured neutral (profiles/r6_ablation.txt section 6), off
#endif
#ifndef DDD_ADAPTIVE_BUTTERFLY
#define DDD_ADAPTIVE_BUTTERFLY 1  // ds_bpermute at lane ^ m directly instead of __shfl_xor
#endif
template <int kRows>
constexpr int adaptive_lean() {
  return ((kRows == 64 ? 1 : 2) & DDD_ADAPTIVE_LEAN) != 0 ? 1 : (DDD_ADAPTIVE_TRIG_LDS ? 2 : 0);
}

template <int kRows, int kWR, bool kHoist, int kEq, bool kWide = false, class TW = DefaultTower>
__global__ __launch_bounds__(kRows / kWR * 64, (min_waves<kRows, kWR, TW, true>())) void adaptive_kernel(
    DevParams p, AdaptiveArgs a) {
  __shared__ Shared<kRows, kWR, kWide, TW> sm;
  __shared__ AdaptiveShared<kRows> as;
  static_assert(kRows == kWR || kWide || !TW::kDefault ||
                    sizeof(Shared<kRows, kWR, kWide>) + sizeof(AdaptiveShared<kRows>) <= 80 * 1024,
                "two 256-row workgroups per CU");
  // reduction scratch of sample_sum: one-wave groups shuffle; four-wave groups use
  // Shared::un + Shared::flux (2 x kRows floats = kRows doubles), free between evaluations
  double* red = kRows == kWR ? nullptr : reinterpret_cast<double*>(sm.un);
  const int tid = (int)threadIdx.x;
  const Lane ln = make_lane<kRows, kWR>(p, a.batch, tid, (int)blockIdx.x);
  Resident res;
  // (no resident tower weights here: the controller state needs the registers -- with them
  // the streamed-tower adaptive kernels spill 3-16 VGPRs)
  const bool fast_frc = launch_setup<kRows, kWR, kHoist, false>(p, sm, ln, a.batch, res);
  // sample whose (sample, mode) pair this lane evaluates in forcing phase 1
  const int frc_sl = (fast_frc && tid < (kRows / p.N) * p.P)
                         ? row_sample(tid, p.inv_P) : 0;
  const bool row_live = ln.row < ln.rows_used;

  const double t0 = a.times[0];
  const double t_bound = a.times[a.n_times - 1];
  const double interval = fabs(t_bound - t0);
  const double rtol = a.rtol, atol = a.atol, max_step = a.max_step;
  const double sqrt_n = sqrt((double)p.N);   // _ivp.common.norm: ||x|| / x.size ** 0.5
  // N a power of four (16, 64, 256): x.size ** 0.5 is a power of two and the division is
  // an exact multiplication -- a dozen float64 instructions less per error norm
  const bool sqrt_n_pow2 = (p.N & (p.N - 1)) == 0 && (__builtin_ctz(p.N) & 1) == 0;
  const double inv_sqrt_n = 1.0 / sqrt_n;
  const size_t row_stride = (size_t)a.batch * p.N;
  const auto rms = [&](double q) {
    const double nrm = sqrt(sample_sum<kRows, kWR>(p, ln, q * q, red));
    return sqrt_n_pow2 ? nrm * inv_sqrt_n : nrm / sqrt_n;
  };
  // forcing sums computed outside an evaluation (the first stage of every attempt): the
  // masked first trip of forcing_phase2 where the evaluation keeps the masks resident anyway
  constexpr bool kMaskedSums = kHoist && kWR == 64 && kEq >= 0 && adaptive_lean<kRows>() != 1 &&
                               spec_folded(kEq >= 0 ? kEq : 0);

  // The step-size controllers live in LDS, one per sample of the group: their 7
  // doubles + 4 ints are identical on all lanes of a sample and are touched once
  // per evaluation, so as per-lane registers they were spilled to scratch (~90
  // scratch operations per attempt at 256 VGPRs).  A lane reads its sample's
  // controller (broadcast reads) where the evaluation's input is formed and again
  // after the evaluation; the lane of grid point 0 writes it back.  Per lane and
  // across the evaluation only the state and the stage derivatives stay live.
  PackedControl* ctl = as.ctl;
  int* attempts_of = as.attempts_of;
  int* vote = as.vote;
  if (tid == 0) { vote[0] = 0; vote[1] = 0; }
  const int slot = ln.sl;
  const bool keeper = row_live && ln.pos == 0 && ln.owner;
  {
    rk23::Control c0;
    c0.init(t0, ln.valid != 0);
    if (keeper) { ctl[slot].store(c0); attempts_of[slot] = 0; }
  }
  group_barrier<kRows, kWR>();
  double y = ln.valid ? a.y0[ln.gidx] : 0.0;
  double y_new = y;
  float k0 = 0.0f, k1 = 0.0f, k2 = 0.0f;
  double te_next = 0.0;   // times[ti] of the lane's sample, requested one evaluation ahead

  // phase 0: f(t0, y0); 1: the probe of select_initial_step; 2..4: stages 2, 3
  // and the FSAL stage of one attempt.  Uniform over the workgroup.
  // select_initial_step's h0 and d1 live only until the first attempt starts:
  // they ride in Control::h and Control::t_new (set by begin_attempt after their
  // last use).
  int phase = 0;
  int round = 0;
  bool sums_ready = false;   // res.fk_next already holds the forcing sums of this evaluation
  // One-wave groups also LOOK AHEAD across the error test (round 6): the FSAL stage's evaluation
  // prepares the sums of the NEXT attempt's first stage for the time that attempt will have if
  // this one is accepted with the controller saturated at max_step (the Burgers case: every
  // attempt) -- checked against the real time once the controller has decided, recomputed when
  // the guess was wrong (a rejection, a step below max_step).  The same instructions on the
  // same operands as the sums computed in place: the same bits.
  constexpr bool kSpeculate = kRows == kWR && DDD_ADAPTIVE_SPECULATE;
  bool sums_guessed = false;
  float guess_lane = 0.0f;   // the time this lane's (sample, mode) pair was prepared for
  DDD_ADAPT_TRACE_SETUP;
  for (;;) {
    DDD_ADAPT_STAMP(0);
    double tt, yy, tt_next;
    {
      const double ct = ctl[slot].t, ch = ctl[slot].h;   // (phase 1: h = h0)
      const int cstatus = ctl[slot].status();
      if (phase == 0) { tt = ct; yy = y; tt_next = tt; }
      else if (phase == 1) { tt = ct + ch; yy = y + ch * (double)k0; tt_next = tt; }
      else if (phase == 2) {
        tt = ct + 0.5 * ch; yy = rk23::stage2_input(y, k0, ch); tt_next = ct + 0.75 * ch;
      } else if (phase == 3) {
        tt = ct + 0.75 * ch; yy = rk23::stage3_input(y, k0, k1, ch); tt_next = ct + ch;
      } else { tt = ct + ch; yy = y_new; tt_next = tt; }
      if (cstatus != rk23::RUNNING) { tt = ct; yy = y; tt_next = ct; }   // idle samples stay finite
    }

    // Harmonic forcing sums at THIS sample's time.  Inside an attempt the next
    // stage's time is known, so stages 3 and 4 get their sums from the look-ahead
    // of the evaluation before them (as the fixed-step integrators do); the first
    // stage of an attempt cannot: its time depends on the error test.
    const bool ahead = fast_frc && (phase == 2 || phase == 3);
    bool prepare = ahead;
    float tn_lane = (float)tt;
    if (fast_frc) {
      // the lane's (sample, mode) pair of forcing phase 1 belongs to sample frc_sl of
      // the group, not to the lane's own: that sample's evaluation times, formed from
      // ITS controller exactly as above
      const double ft = ctl[frc_sl].t, fh = ctl[frc_sl].h;
      const bool frun = ctl[frc_sl].status() == rk23::RUNNING;
      double ft_now = ft, ft_next = ft;
      if (frun) {
        if (phase == 1) { ft_now = ft + fh; ft_next = ft_now; }
        else if (phase == 2) { ft_now = ft + 0.5 * fh; ft_next = ft + 0.75 * fh; }
        else if (phase == 3) { ft_now = ft + 0.75 * fh; ft_next = ft + fh; }
        else if (phase == 4) { ft_now = ft + fh; ft_next = ft_now; }
      }
      if (kSpeculate && sums_guessed) {   // (wave-uniform)
        const bool wrong = tid < res.frc_pairs && (float)ft_now != guess_lane;
        sums_ready = __builtin_amdgcn_ballot_w64(wrong) == 0ull;
        sums_guessed = false;
      }
      if (!sums_ready)
        res.fk_next = forcing_sums<kRows, kWR, kMaskedSums>(p, sm, res, (float)ft_now, tid);
      tn_lane = (float)ft_next;
      if (kSpeculate && phase == 4) {
        // the next attempt of sample frc_sl as Control::advance / begin_step / begin_attempt
        // will set it up if this one is accepted at h_abs = max_step: t = t_new,
        // t_new' = min(t + max_step, t_bound), h = t_new' - t; its first stage at t + h / 2
        const double tp = ctl[frc_sl].t_new;
        double tq = tp + max_step;
        if (tq - t_bound > 0.0) tq = t_bound;
        const double guess = frun ? tp + 0.5 * (tq - tp) : ft;
        tn_lane = (float)guess;
        guess_lane = tn_lane;
        prepare = true;
        sums_guessed = true;
      }
    }
    DDD_ADAPT_STAMP(1);
    const float f = eval_rhs<kRows, kWR, kHoist, kEq, false, kWide, adaptive_lean<kRows>(), TW>(
        p, sm, a.batch, (float)yy, (float)tt, tn_lane, res, fast_frc, nullptr, nullptr, prepare);
    sums_ready = ahead;
    DDD_ADAPT_STAMP(2);
    if (DDD_ADAPTIVE_SHORTCUT && (phase == 2 || phase == 3)) {
      // Stages 2 and 3 of an attempt change nothing of the controller but the evaluation
      // count, and no sample can finish here: no controller load / unpack / store, no vote
      // (two of every three evaluations).  Stage 3 also requests the next output time: the
      // dense-output loop after stage 4 used to wait a global-memory round trip for it in
      // every attempt.
      const bool run = ctl[slot].status() == rk23::RUNNING;
      if (phase == 2) {
        k1 = f;
        phase = 3;
      } else {
        k2 = f;
        y_new = rk23::new_state(y, k0, k1, k2, ctl[slot].h);
        if (DDD_ADAPTIVE_PREFETCH) {
          const int ti = ctl[slot].bits >> 3;
          te_next = a.times[ti < a.n_times ? ti : a.n_times - 1];
        }
        phase = 4;
      }
      if (keeper && run) ctl[slot].nfev += 1;   // (the keeper alone touches this field)
      DDD_ADAPT_STAMP(3);
      DDD_ADAPT_STAMP(4);
      continue;
    }
    // (eval_rhs's barriers are compiler barriers too: this is a fresh read)
    rk23::Control c = ctl[slot].load();
    if (c.status == rk23::RUNNING) ++c.nfev;
    int attempts = -1;   // phase 4: the sample's attempt count after this one
    double& h0 = c.h;
    double& d1 = c.t_new;

    if (phase == 0) {
      k0 = f;
      if (a.n_times == 1) {   // t0 == t_bound: nothing to integrate
        if (c.status == rk23::RUNNING) {
          if (ln.active) a.y_out[ln.gidx] = y;
          c.ti = 1; c.status = rk23::FINISHED;
        }
      } else {
        const double scale = atol + fabs(y) * rtol;
        const double d0 = rms(y / scale);
        d1 = rms((double)k0 / scale);
        h0 = rk23::Control::first_guess(d0, d1, interval);
      }
      phase = 1;
    } else if (phase == 1) {
      const double scale = atol + fabs(y) * rtol;
      const double d2 = rms((double)(f - k0) / scale) / h0;   // float32 difference, as SciPy forms it
      c.initial_step(h0, d1, d2, interval, max_step);
      c.begin_step(max_step);
      c.begin_attempt(t_bound);
      phase = 2;
    } else if (!DDD_ADAPTIVE_SHORTCUT && phase == 2) {
      k1 = f;
      phase = 3;
    } else if (!DDD_ADAPTIVE_SHORTCUT && phase == 3) {
      k2 = f;
      y_new = rk23::new_state(y, k0, k1, k2, c.h);
      phase = 4;
    } else {   // phase 4 (2 and 3: above)
      const float k3 = f;
      const double error_norm =
          rms(rk23::scaled_error(y, y_new, k0, k1, k2, k3, c.h, rtol, atol));
      if (c.status == rk23::RUNNING && c.error_test(error_norm)) {
        // solve_ivp: dense output at every t_eval in (t_old, t_new]
        // (times[c.ti] was requested during stage 3: te_next)
        double te = DDD_ADAPTIVE_PREFETCH ? te_next : a.times[c.ti < a.n_times ? c.ti : a.n_times - 1];
        while (c.ti < a.n_times) {
          if (!(te <= c.t_new)) break;
          // (spare rows of a 256-row group follow sample 0's controller: no stores)
          if (ln.active)
            a.y_out[(size_t)c.ti * row_stride + ln.gidx] =
                rk23::dense_output(y, k0, k1, k2, k3, (te - c.t) / c.h, c.h);
          ++c.ti;
          if (c.ti < a.n_times) te = a.times[c.ti];
        }
        y = y_new;
        k0 = k3;
        c.advance(t_bound, max_step);
      }
      attempts = attempts_of[slot];
      if (attempts < 0x7fffffff) ++attempts;
      if (c.status == rk23::RUNNING && (long long)attempts >= a.max_attempts)
        c.status = rk23::ATTEMPT_LIMIT;
      c.begin_attempt(t_bound);
      phase = 2;
    }
    // every lane of the sample has read the controller (above); the keeper publishes
    // the new one.  The vote below is the barrier that orders it before the next read.
    const int running = c.status == rk23::RUNNING;
    DDD_ADAPT_STAMP(3);
    if (kRows != kWR) __syncthreads();   // four-wave groups: all lanes' reads before the write
    if (keeper) {
      ctl[slot].store(c);
      if (attempts >= 0) attempts_of[slot] = attempts;
    }
    if (kRows == kWR) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!__any(running)) break;
    } else {
      // workgroup vote through two alternating LDS flags.  (Not __syncthreads_or:
      // its device-library reduction makes the compiler assume the kernel may need
      // AGPRs, the MFMAs then take their AGPR form and 256 VGPRs + 72 AGPRs leave
      // ONE wavefront per SIMD instead of two: KS N = 256 ran at 49 % of peak.)
      if (running) vote[round & 1] = 1;
      if (tid == 0) vote[(round + 1) & 1] = 0;   // next round's flag; its readers are long gone
      __syncthreads();
      const int go = vote[round & 1];
      ++round;
      if (!go) break;
    }
    DDD_ADAPT_STAMP(4);
  }

  const rk23::Control c = ctl[slot].load();
  if (ln.active) {
    if (c.status != rk23::FINISHED) {
      const double nan = __longlong_as_double(0x7ff8000000000000ll);
      for (int i = c.ti; i < a.n_times; ++i) a.y_out[(size_t)i * row_stride + ln.gidx] = nan;
    }
    if (ln.pos == 0) {
      const long sample = ln.gidx / p.N;
      a.nfev[sample] = c.nfev;
      a.status[sample] = c.status;
    }
  }
}

}  // namespace mfma
}  // namespace ddd
