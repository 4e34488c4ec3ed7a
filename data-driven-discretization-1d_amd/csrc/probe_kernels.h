// Hardware probes behind profiles/tools/ (placement, MFMA issue rates, VALU /
// MFMA issue sharing).  Compiled into libddd1d_probe.so only (-DDDD_PROBES,
// __graft_entry__.build_probe): never part of the product library.
#pragma once
#include <hip/hip_runtime.h>

#include "ops.h"   // f32x4 / f32x16, mfma4_bcast

namespace ddd {
namespace ops {

// Debug: record the hardware placement of each single-wave workgroup
// (HW_REG_HW_ID and XCC_ID) under the same LDS footprint as the 64-row kernel.
__global__ __launch_bounds__(64) void hwid_probe_kernel(unsigned* __restrict__ out, int spin) {
  __shared__ float pad[5088];   // 20352 B, as Shared<64>
  pad[threadIdx.x] = (float)blockIdx.x;
  unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_ID, 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // XCC_ID
  float acc = pad[threadIdx.x];
  for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;       // keep waves resident
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = hw;
    out[blockIdx.x * 4 + 1] = xcc;
    out[blockIdx.x * 4 + 2] = (unsigned)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 4 + 3] = __float_as_uint(acc);
  }
}

// Debug: matrix-pipe rate probe.  Each wave issues `iters` x 8 MFMAs in
// `chains` (1, 2 or 4) independent accumulator chains and reports s_memtime
// ticks, so ticks per MFMA can be compared with the nominal 64 / 32 cycles.
template <int kChains>
__global__ __launch_bounds__(64) void mfma4_rate_probe_kernel(unsigned long long* out,
                                                               int iters, float seed) {
  f32x4 acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 4; ++r) acc[c][r] = seed * (float)(c + r);
  const float x = seed + threadIdx.x, y = seed * 0.5f - threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 24; ++k) {
      const int c = k % kChains;
      if ((k / kChains) & 1) acc[c] = mfma4_bcast<5>(x, y, acc[c]);
      else acc[c] = mfma4_bcast<10>(x, y, acc[c]);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sink = 0.0f;
  for (int c = 0; c < 4; ++c) sink += acc[c][0] + acc[c][3];
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2 + 0] = t1 - t0;
    out[blockIdx.x * 2 + 1] = (unsigned long long)__float_as_uint(sink);
  }
}

// Debug: how two wavefronts of one SIMD share the issue ports.  Workgroups of
// eight wavefronts (two per SIMD: the LDS block admits one workgroup per CU,
// the register budget two wavefronts per SIMD).  Hardware wave slot 0 of each
// SIMD streams MFMAs (kind 0: none / idle, 1: 32x32x2 two chains, 2: 4x4x1
// three chains, 3-6 / 7-8: the same streams paced with s_nops), slot 1 runs `work` (0: idle, 1: dependent v_fma chain, 2:
// independent v_fma x4, 3: LDS read-modify chain, 4: the token poll loop shape:
// ds_read + readfirstlane + s_sleep).  Each reports s_memtime ticks for
// `iters` x 64 operations; prio: s_setprio of the worker.
template <int kMfma, int kWork>
__global__ __launch_bounds__(512, 1) void issue_share_probe_kernel(unsigned long long* out,
                                                                   int iters, float seed,
                                                                   int prio) {
  __shared__ float pad[40000];   // 160000 B: one workgroup per CU
  const int lane = threadIdx.x & 63;
  pad[threadIdx.x] = seed;
  __syncthreads();
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 0xfu;   // HW_ID.wave_id
  const unsigned simd = __builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4);
  unsigned long long ticks = 0;
  float sink = 0.0f;
  if (slot == 0) {
    f32x16 a32[2];
    f32x4 a4[3];
    for (int c = 0; c < 2; ++c) for (int r = 0; r < 16; ++r) a32[c][r] = seed * (float)(c + r);
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 4; ++r) a4[c][r] = seed * (float)(c + r);
    const float x = seed + lane, y = seed * 0.5f - lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (kMfma == 1) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k)
          a32[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a32[k & 1], 0, 0, 0);
      }
    } else if (kMfma >= 3 && kMfma <= 6) {
      // paced stream: s_nops after each MFMA keep this wavefront's NEXT MFMA out
      // of the issue stage while the matrix pipe is busy (3: 16, 4: 32, 5: 48,
      // 6: 56 idle cycles per 64-cycle MFMA)
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
          asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(a32[k & 1]) : "v"(x), "v"(y));
          asm volatile("s_nop 15");
          if (kMfma >= 4) asm volatile("s_nop 15");
          if (kMfma >= 5) asm volatile("s_nop 15");
          if (kMfma >= 6) asm volatile("s_nop 7");
        }
      }
    } else if (kMfma == 7 || kMfma == 8) {
      // paced 4x4x1 stream (7: s_nop 1 = 2 idle cycles, 8: s_nop 3 = 4 idle cycles per MFMA)
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 256; ++k) {
          asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0 cbsz:4 abid:5" : "+v"(a4[k % 3]) : "v"(x), "v"(y));
          if (kMfma == 7) asm volatile("s_nop 1"); else asm volatile("s_nop 3");
        }
      }
    } else if (kMfma == 2) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) a4[k % 3] = mfma4_bcast<3>(x, y, a4[k % 3]);
#pragma unroll
        for (int k = 0; k < 64; ++k) a4[k % 3] = mfma4_bcast<7>(x, y, a4[k % 3]);
#pragma unroll
        for (int k = 0; k < 64; ++k) a4[k % 3] = mfma4_bcast<9>(x, y, a4[k % 3]);
#pragma unroll
        for (int k = 0; k < 64; ++k) a4[k % 3] = mfma4_bcast<12>(x, y, a4[k % 3]);
      }
    }
    ticks = __builtin_amdgcn_s_memtime() - t0;
    sink = a32[0][0] + a32[1][3] + a4[0][0] + a4[1][1] + a4[2][2];
  } else {
    if (prio) __builtin_amdgcn_s_setprio(3);
    float v0 = seed + lane, v1 = seed - lane, v2 = seed * 2.0f, v3 = seed * 3.0f;
    const float m = 1.0001f, c = 0.5f;
    volatile float* lp = pad + 1024 + (threadIdx.x & 511);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (kWork == 1) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(m), "v"(c));
      }
    } else if (kWork == 2) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(m), "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v1) : "v"(m), "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v2) : "v"(m), "v"(c));
          asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(m), "v"(c));
        }
      }
    } else if (kWork == 3) {
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) { const float q = *lp; *lp = q + c; }
      }
    } else if (kWork == 4) {
      int acc = 0;
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 64; ++k) {
          acc += __builtin_amdgcn_readfirstlane(*(volatile int*)lp);
          __builtin_amdgcn_s_sleep(1);
        }
      }
      v0 = (float)acc;
    }
    ticks = __builtin_amdgcn_s_memtime() - t0;
    sink = v0 + v1 + v2 + v3;
  }
  if (lane == 0) {
    const int idx = (blockIdx.x * 8 + (threadIdx.x >> 6)) * 4;
    out[idx + 0] = ticks;
    out[idx + 1] = slot | (simd << 8);
    out[idx + 2] = (unsigned long long)__float_as_uint(sink);
    out[idx + 3] = 0;
  }
}

template <int kChains, bool k32>
__global__ __launch_bounds__(64) void mfma_rate_probe_kernel(unsigned long long* out,
                                                              int iters, float seed) {
  f32x16 a32[4];
  f32x4 a16[4];
  for (int c = 0; c < 4; ++c) {
    for (int r = 0; r < 16; ++r) a32[c][r] = seed * (float)(c + r);
    for (int r = 0; r < 4; ++r) a16[c][r] = seed * (float)(c + r);
  }
  const float x = seed + threadIdx.x, y = seed * 0.5f - threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int c = k % kChains;
      if (k32) a32[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a32[c], 0, 0, 0);
      else a16[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a16[c], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float sink = 0.0f;
  for (int c = 0; c < 4; ++c) sink += a32[c][0] + a16[c][0];
  if (threadIdx.x == 0) {
    out[blockIdx.x * 2 + 0] = t1 - t0;
    out[blockIdx.x * 2 + 1] = (unsigned long long)__float_as_uint(sink);
  }
}

}  // namespace ops
}  // namespace ddd
