// Phase stamps of mfma::adaptive_kernel -- libddd1d_probe.so only (-DDDD_PROBES; never part of
// the product library: rhs_adaptive.h includes this file under that macro and compiles the
// hooks to nothing otherwise).  profiles/tools/adaptive_phase_trace.py.
//
// No field is added to AdaptiveArgs (its layout is the product's): the tool asks for tracing
// by setting bit 62 of `max_attempts` (an attempt limit no run reaches) and hands an `nfev`
// buffer that is followed, at the next 8-byte boundary after `batch` ints, by
// [blocks][kTraceSlots] 64-bit words.  Thread 0 of every workgroup stamps the first 48 loop
// iterations, five stamps each: (s_memtime << 3) | phase-at-loop-top at
//   0 loop top   1 before eval_rhs (inputs formed, first-stage forcing sums done)
//   2 after eval_rhs   3 controller arithmetic done (error norm, error test, dense output)
//   4 controller published, workgroup vote taken.
#pragma once
#define DDD_ADAPT_TRACE_SETUP                                                                    \
  unsigned long long* adapt_trace = nullptr;                                                     \
  int adapt_iter = 0, adapt_phase = 0;                                                           \
  if ((a.max_attempts >> 62) & 1)                                                                \
    adapt_trace = reinterpret_cast<unsigned long long*>(a.nfev + ((a.batch + 1) & ~1)) +         \
                  (size_t)blockIdx.x * ddd::kTraceSlots
#define DDD_ADAPT_STAMP(K)                                                                       \
  do {                                                                                           \
    if ((K) == 0) adapt_phase = phase;                                                           \
    if (adapt_trace != nullptr && threadIdx.x == 0 && adapt_iter < 48)                           \
      adapt_trace[adapt_iter * 5 + (K)] =                                                        \
          ((unsigned long long)__builtin_amdgcn_s_memtime() << 3) | (unsigned)adapt_phase;       \
    if ((K) == 4) ++adapt_iter;                                                                  \
  } while (0)
