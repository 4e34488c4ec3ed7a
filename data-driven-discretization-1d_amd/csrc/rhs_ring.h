// The command-ring substep kernel (kernel 1r): rhs_mfma.h's multi-group substep walk under a
// device-resident command ring.  Its own header so that work on it recompiles the six
// mfma_ring units only.
#pragma once
#include "rhs_mfma.h"
#include "ring_args.h"

namespace ddd {
namespace mfma {

// Kernel 1r: the same walk under a device-resident command ring (ring_args.h).
// One launch serves every ddd_rk_substep call of a ddd_stream_fork .. ddd_stream_join
// region: the group sets its weights up once, then takes command after command.  One-wave
// groups only (N | 64): a group is a wavefront, the command is fetched by its first
// kRingChunks lanes and broadcast by v_readlane -- the arguments sit in SGPRs exactly like a
// kernel-argument segment.  The arithmetic of a command is substep_walk's: the same bits as
// one launch per substep.
//
// MEASURED (profiles/r6_ablation.txt, r6_ring_trace.txt): 31.6 us per call at 4 096 samples =
// 71.6 % -- the rate of the launches it replaces (31.8 us), not the persistent integrator's
// 27.5 us.  What a substep boundary costs is not the launch: every command starts cold --
// arguments, state, forcing rows, harmonic sums in front of the first MFMA, two or three
// memory round trips that the SIMD partner's evaluation stretches to 8 us each -- and that
// is the same under a launch and under a command.  Tried on top of this kernel, all measured
// SLOWER (kept out): the next command's slot requested one command ahead straight into LDS
// (global_load_lds) with its harmonic sums prepared inside the current command's last
// evaluation and its state handed over in registers (34.5 us: every command "warm", 216 bytes
// of scratch); a rotating wavefront relaying eight slots ahead (no change); the bookkeeping
// at raised issue priority (s_setprio 3 outside the evaluations: the 8-us segments shrink to
// 1.6 us and the time reappears in the command fetch -- 32.8 us either way).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// -DDDD_RING_TRACE=1 (a variant build, profiles/tools/ring_trace.py): s_memrealtime stamps
// (100 MHz) of the first commands of groups 0 .. 7, in the page-locked block behind
// RingArgs::status
#ifdef DDD_RING_TRACE
#ifndef DDD_RING_TRACE_SKIP
#define DDD_RING_TRACE_SKIP 120   // commands of a launch before the ten that are stamped
#endif
// stamp k (0 .. 11) of the current command (ncmd: commands this launch has taken)
#define DDD_RING_STAMP(k) do { if (first < 8 && tid == 0 && ncmd >= DDD_RING_TRACE_SKIP && ncmd < DDD_RING_TRACE_SKIP + 10) \
    reinterpret_cast<unsigned long long*>(r.status + 16)[first * 128 + (ncmd - DDD_RING_TRACE_SKIP) * 12 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define DDD_RING_STAMP(k) do { } while (0)
#endif

// lanes 0 .. kRingChunks - 1 hold a command's chunks: its arguments as wave-uniform values
__device__ __forceinline__ void ring_decode(const u32x4& c, SubstepArgs& a) {
  const auto dw = [&](int j) -> unsigned {
    const unsigned v = (j % 3 == 0) ? c.x : (j % 3 == 1) ? c.y : c.z;
    return (unsigned)__builtin_amdgcn_readlane((int)v, j / 3);
  };
  const auto qw = [&](int j) -> unsigned long long {
    return (unsigned long long)dw(j) | ((unsigned long long)dw(j + 1) << 32);
  };
  a.t = __longlong_as_double((long long)qw(0));
  a.y_in = reinterpret_cast<const float*>(qw(2));
  a.y_base = reinterpret_cast<const float*>(qw(4));
  a.y_out = reinterpret_cast<float*>(qw(6));
  a.acc_in = reinterpret_cast<const float*>(qw(8));
  a.acc_out = reinterpret_cast<float*>(qw(10));
  a.c1 = __uint_as_float(dw(12));
  a.c2 = __uint_as_float(dw(13));
  a.batch = (int)dw(14);
  a.derivs_out = nullptr;
  a.coeffs_out = nullptr;
}

// One relay pass, if the lock is free: host slots from .. from + 7 in one wave-wide load
// (lane = 8 slot + chunk), the posted ones copied to the device ring.  Returns false when
// another wavefront holds the lock; h: this lane's chunk, good: lanes whose chunk is posted
// (or unused).
__device__ __forceinline__ bool ring_relay(const RingArgs& r, unsigned from, int lane, u32x4& h,
                                           unsigned long long& good) {
  unsigned* lock = r.count + kRingSlots;
  int got = 0;
  if (lane == 0)
    got = __hip_atomic_exchange(lock, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u;
  got = __builtin_amdgcn_readfirstlane(got);
  if (!got) return false;
  const unsigned j = (unsigned)lane >> 3, ch = (unsigned)lane & 7u;
  const unsigned slj = (from + j) & (kRingSlots - 1);
  h = reinterpret_cast<const volatile u32x4*>(r.slots)[(size_t)slj * kRingSlotChunks + ch];
  // slot j is posted when its chunks in use carry the tag of from + j
  good = __builtin_amdgcn_ballot_w64(ch >= (unsigned)kRingChunks || h.w == from + 1u + j);
  const bool posted = ((good >> (8 * j)) & 0xffull) == 0xffull;
  if (posted && ch < (unsigned)kRingChunks)
    reinterpret_cast<volatile u32x4*>(r.dev_slots)[(size_t)slj * kRingSlotChunks + ch] = h;
  if (lane == 0)   // (release: the copies above are out before the lock opens)
    __hip_atomic_store(lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// Wait for command `index`; false: the watchdog expired (RingArgs::status set).
template <int kRows, int kWR>
__device__ __forceinline__ bool ring_take(const RingArgs& r, unsigned index, int lane,
                                          SubstepArgs& a) {
  static_assert(kRows == kWR, "command ring: one-wave groups");
  const unsigned tag = index + 1u;
  const volatile u32x4* dev = reinterpret_cast<const volatile u32x4*>(r.dev_slots) +
                              (size_t)(index & (kRingSlots - 1)) * kRingSlotChunks;
  u32x4 c = {0u, 0u, 0u, tag};
  unsigned long long t0 = 0;
  for (int spins = 0;; ++spins) {
    if (lane < kRingChunks) c = dev[lane];   // (volatile: a system-scope load, never a cached copy)
    if (__builtin_amdgcn_ballot_w64(c.w != tag) == 0ull) break;
    // not relayed yet: the first wavefront to get the lock reads the host ring and copies what
    // is posted; the others look at the device ring again
    u32x4 h = {0u, 0u, 0u, 0u};
    unsigned long long good = 0ull;
    const bool got = ring_relay(r, index, lane, h, good);
    if (got && (good & 0xffull) == 0xffull) {   // our own command was among them: lanes 0..5 hold it
      if (lane < kRingChunks) c = h;
      break;
    }
    if (got) __builtin_amdgcn_s_sleep(32);   // (the host has not posted it: look again in ~1 us)
    else __builtin_amdgcn_s_sleep(8);
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    if (spins == 0) t0 = now;
    if (now - t0 > (unsigned long long)r.watchdog_ticks) {
      if (lane == 0)
        __hip_atomic_store(r.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
  }
  ring_decode(c, a);
  return true;
}

template <int kRows, int kWR, int kEq>
__global__ __launch_bounds__(kRows / kWR * 64, 2) void substep_ring_kernel(DevParams p, RingArgs r) {
  __shared__ Shared<kRows, kWR> sm;
  const int tid = group_tid<kRows, kWR>();
  const int first = (int)blockIdx.x, stride = (int)gridDim.x;
  const int spg = kRows / p.N;
  Resident res;
  {
    const Lane ln0 = make_lane<kRows, kWR>(p, 0, tid, first);   // (the setup reads lane geometry only)
    setup_weights<kRows, kWR, true>(p, sm, ln0, res);
  }
  int ncmd = 0;   // (trace builds only)
  (void)ncmd;
  const bool fast_frc = forcing_is_fast<kRows, kWR>(p);
  // groups that had passed the previous command before this one (its fetch-and-add, looked
  // at one command later: the answer is back by then, nothing waits for it)
  unsigned passed = 0u;
  for (unsigned index = r.first_index;; ++index) {
    SubstepArgs a;
    const bool taken = ring_take<kRows, kWR>(r, index, tid, a);
    DDD_RING_STAMP(0);   // command known
    // the previous command is done with when every group of the launch has passed it: the
    // last one tells the host, which may then reuse the slot (commands are never re-read)
    if (tid == 0 && index != r.first_index && passed + 1u == (unsigned)stride) {
      const unsigned sl = (index - 1u) & (kRingSlots - 1);
      __hip_atomic_store(r.count + sl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(r.done + sl, index, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (!taken) return;
    if (a.batch == kRingStop) {
      // the stop command's slot is reported like any other (the host's back-pressure spans
      // launches: the next kernel's commands must not overwrite what this one still reads)
      if (tid == 0) {
        const unsigned sl = index & (kRingSlots - 1);
        const unsigned before = __hip_atomic_fetch_add(r.count + sl, 1u, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        if (before + 1u == (unsigned)stride) {
          __hip_atomic_store(r.count + sl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(r.done + sl, index + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      return;
    }
    const int groups = (a.batch + spg - 1) / spg;
    if (first < groups) {
      Lane ln = make_lane<kRows, kWR>(p, a.batch, tid, first);
      float u = ln.valid ? a.y_in[ln.gidx] : 0.0f;
      apply_samples<kRows, kWR>(sm, res, fetch_samples<kRows, kWR>(p, first, a.batch, fast_frc));
      if (fast_frc) res.fk_next = forcing_sums<kRows, kWR, true>(p, sm, res, (float)a.t, tid);
      DDD_RING_STAMP(1);   // first group's state and harmonic sums there
      for (int grp = first; grp < groups; grp += stride) {
        // (the body of substep_walk's loop)
        const int nxt = grp + stride;
        const bool more = nxt < groups;
        Lane ln_next = ln;
        float u_next = 0.0f;
        SampleSetup s_next{0.0f, 0.0f, 0.0f, 0, 0, 0};
        if (more) {
          ln_next = make_lane<kRows, kWR>(p, a.batch, tid, nxt);
          u_next = ln_next.valid ? a.y_in[ln_next.gidx] : 0.0f;
          s_next = fetch_samples<kRows, kWR>(p, nxt, a.batch, fast_frc);
        }
        const float base = (ln.active && a.y_out != nullptr && a.y_base != nullptr)
                               ? a.y_base[ln.gidx] : 0.0f;
        const float acc_in = (ln.active && a.acc_out != nullptr && a.acc_in != nullptr)
                                 ? a.acc_in[ln.gidx] : 0.0f;
        if (more) apply_samples<kRows, kWR, false>(sm, res, s_next);
        DDD_RING_STAMP(grp == first ? 2 : 6);   // evaluation starts
        const float f = eval_rhs<kRows, kWR, true, kEq, false>(p, sm, a.batch, u, (float)a.t,
                                                               (float)a.t, res, fast_frc, nullptr,
                                                               nullptr, more, grp);
        DDD_RING_STAMP(grp == first ? 3 : 7);   // evaluation done
        if (ln.active) {
          if (a.y_out != nullptr) a.y_out[ln.gidx] = base + a.c1 * f;
          if (a.acc_out != nullptr) a.acc_out[ln.gidx] = acc_in + a.c2 * f;
        }
        DDD_RING_STAMP(grp == first ? 4 : 8);   // results stored (requests out)
        group_barrier<kRows, kWR>();   // this group's epilogue has read sm.fk / sm.u
        DDD_RING_STAMP(grp == first ? 5 : 9);
        ln = ln_next;
        u = u_next;
      }
    }
    if (tid == 0)
      passed = __hip_atomic_fetch_add(r.count + (index & (kRingSlots - 1)), 1u, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
    DDD_RING_STAMP(10);   // counted
    ++ncmd;
  }
}

}  // namespace mfma
}  // namespace ddd
