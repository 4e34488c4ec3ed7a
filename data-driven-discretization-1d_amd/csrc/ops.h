// Standalone operators: the reference's unit-test surface for this path
// (layers_test.py, polynomials_test.py) plus the MFMA register-layout probe.
#pragma once
#include "dev_params.h"

namespace ddd {
namespace ops {

// layers.nn_conv1d_periodic / conv1d_periodic_layer (layers.py:95-137):
// out[b,x,f] = bias[f] + sum_{k,c} in[b, (x + k - left) mod N, c] * w[k,c,f],
// left = ceil((K-1)/2) when centred, 0 otherwise (layers.py:76-83).
__global__ void conv1d_periodic_kernel(const float* __restrict__ in,
                                       const float* __restrict__ w,
                                       const float* __restrict__ bias,
                                       float* __restrict__ out, int batch, int n,
                                       int cin, int cout, int k, int left, int act) {
  const long total = (long)batch * n * cout;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int co = (int)(idx % cout);
    const long bx = idx / cout;
    const int x = (int)(bx % n);
    const long b = bx / n;
    float acc = 0.0f;
    for (int kk = 0; kk < k; ++kk) {
      int src = (x + kk - left) % n;
      src = src < 0 ? src + n : src;
      const float* row = in + ((size_t)b * n + src) * cin;
      const float* wk = w + (size_t)kk * cin * cout + co;
      for (int ci = 0; ci < cin; ++ci) acc = fmaf(row[ci], wk[(size_t)ci * cout], acc);
    }
    if (bias != nullptr) acc = acc + bias[co];
    out[idx] = apply_activation(acc, act);
  }
}

// layers.pad_periodic (layers.py:39-83): [B][N][C] -> [B][N+padding][C].
__global__ void pad_periodic_kernel(const float* __restrict__ in,
                                    float* __restrict__ out, int batch, int n,
                                    int c, int padding, int left) {
  const int np = n + padding;
  const long total = (long)batch * np * c;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(idx % c);
    const long bx = idx / c;
    const int x = (int)(bx % np);
    const long b = bx / np;
    int src = (x - left) % n;
    src = src < 0 ? src + n : src;
    out[idx] = in[((size_t)b * n + src) * c + ch];
  }
}

// PolynomialAccuracyLayer.apply (polynomials.py:266-277):
// out[m][g] = bias[g] + sum_i in[m][i] * nullspace[i][g].
__global__ void polynomial_accuracy_kernel(const float* __restrict__ in,
                                           const float* __restrict__ nullspace,
                                           const float* __restrict__ bias,
                                           float* __restrict__ out, long m,
                                           int input_size, int g) {
  const long total = m * g;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int gg = (int)(idx % g);
    const long row = idx / g;
    float acc = 0.0f;
    for (int i = 0; i < input_size; ++i)
      acc = fmaf(in[row * input_size + i], nullspace[i * g + gg], acc);
    out[idx] = bias[gg] + acc;
  }
}

// MFMA layout probe: D = A * B with A[i][k] = 1 + i + 100 k (distinct per
// element) and B[k][j] = (k == 0) ? (j == j0) : 0 style one-hot products are
// overkill; instead feed integer-valued matrices whose product identifies
// every (i, j) uniquely and compare against the layout the kernels assume.
//   32x32x2: A[i][k] = (k == 0) ? i + 1 : 0,  B[k][j] = (k == 0) ? 64 (j + 1) : 0
//            => D[i][j] = 64 (i + 1)(j + 1)  (asymmetric in i <-> j via the 64)
//   plus a second product exercising k == 1 so that a k-swap is caught.
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// model.extract_patches (model.py:516-533): out[b][x][i] = in[b][(x + i - size/2) mod N]
// -- pad_periodic(size - 1, center=True) + tf.extract_image_patches.
__global__ void extract_patches_kernel(const float* __restrict__ in, float* __restrict__ out,
                                       int batch, int n, int size) {
  const long total = (long)batch * n * size;
  const int left = size / 2;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const int i = (int)(idx % size);
    const long bx = idx / size;
    const int x = (int)(bx % n);
    const long b = bx / n;
    int src = (x + i - left) % n;
    src = src < 0 ? src + n : src;
    out[idx] = in[b * n + src];
  }
}

// model.apply_coefficients (model.py:536-548):
// out[b][x][d] = sum_i coeff[b][x][d][i] * in[b][(x + i - G/2) mod N]   (einsum 'bxdi,bxi->bxd')
__global__ void apply_coefficients_kernel(const float* __restrict__ coeff,
                                          const float* __restrict__ in,
                                          float* __restrict__ out, int batch, int n, int d,
                                          int g) {
  const long total = (long)batch * n * d;
  const int left = g / 2;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total;
       idx += (long)gridDim.x * blockDim.x) {
    const long bx = idx / d;
    const int x = (int)(bx % n);
    const long b = bx / n;
    const float* __restrict__ c = coeff + idx * g;
    float acc = 0.0f;
    for (int i = 0; i < g; ++i) {
      int src = (x + i - left) % n;
      src = src < 0 ? src + n : src;
      acc = fmaf(c[i], in[b * n + src], acc);
    }
    out[idx] = acc;
  }
}

// model.apply_space_derivatives (model.py:115-135): equation_of_motion on given
// derivatives; flux forms take the staggered difference of the flux
// (equations.py:305-320).  One workgroup per sample, flux staged in LDS.
__global__ void apply_space_derivatives_kernel(const float* __restrict__ derivs,
                                               const float* __restrict__ y,
                                               float* __restrict__ out, int equation, int n,
                                               int d, float eta, float inv_dx, int flux_form) {
  extern __shared__ float flux[];
  const size_t off = (size_t)blockIdx.x * n;
  for (int x = threadIdx.x; x < n; x += blockDim.x) {
    float dv[kMaxDerivs] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int j = 0; j < d && j < kMaxDerivs; ++j) dv[j] = derivs[(off + x) * d + j];
    flux[x] = equation_rhs_or_flux(equation, y[off + x], dv, eta);
  }
  __syncthreads();
  for (int x = threadIdx.x; x < n; x += blockDim.x) {
    const float here = flux[x];
    out[off + x] = flux_form ? -(inv_dx * (flux[x + 1 == n ? 0 : x + 1] - here)) : here;
  }
}

__global__ void dpp_rotate_probe_kernel(float* __restrict__ out) {
  out[threadIdx.x] = wave_rotate_left1((float)(threadIdx.x * 3 + 1));
}

__global__ void mfma_layout_probe_kernel(float* __restrict__ out32,
                                         float* __restrict__ out16) {
  const int l = threadIdx.x;
  {
    // A[i][k]: lane l supplies i = l & 31, k = l >> 5.
    const int i = l & 31, k = l >> 5;
    const float a = (k == 0) ? (float)(i + 1) : (float)(1000 + i);
    // B[k][j]: lane l supplies k = l >> 5, j = l & 31.
    const int j = l & 31;
    const float b = (k == 0) ? (float)(64 * (j + 1)) : (float)(3 * j + 7);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    // Store by the ASSUMED layout: register r of lane l is
    // D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
      out32[row * 32 + (l & 31)] = acc[r];
    }
  }
  {
    const int i = l & 15, k = l >> 4;
    const float a = (float)((k + 1) * 100 + i);
    const int j = l & 15;
    const float b = (float)((k + 2) * (j + 1));
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    // ASSUMED: register r of lane l is D[4 (l >> 4) + r][l & 15].
    for (int r = 0; r < 4; ++r) out16[(4 * (l >> 4) + r) * 16 + (l & 15)] = acc[r];
  }
}

// v_mfma_f32_4x4x1_16b_f32 with A-block broadcast (cbsz = 4: block `abid` of the A
// register feeds all 16 blocks).  ASSUMED layout, checked by ddd_selftest_mfma_layout:
//   A: lane l holds A[i = l & 3] of block l >> 2;  B: lane l holds B[j = l & 3] of block l >> 2;
//   D: register r of lane l is D[i = r][j = l & 3] of block l >> 2
// so with the broadcast, register r of lane l = A(lane 4 abid + r) * B(lane l) + C:
// every lane keeps its OWN column -- four output channels of its own grid point.
template <int kAbid>
__device__ __forceinline__ f32x4 mfma4_bcast(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, kAbid, 0);
}

__global__ void mfma4_layout_probe_kernel(float* __restrict__ out) {   // [16][4][64]
  const int l = threadIdx.x;
  const float a = (float)(1000 * (l >> 2) + (l & 3) + 1);
  const float b = (float)(l + 1);
#define DDD_P4(ABID)                                                        \
  {                                                                         \
    f32x4 acc = {0.5f, 0.5f, 0.5f, 0.5f};                                   \
    acc = mfma4_bcast<ABID>(a, b, acc);                                     \
    for (int r = 0; r < 4; ++r) out[(ABID * 4 + r) * 64 + l] = acc[r];      \
  }
  DDD_P4(0) DDD_P4(1) DDD_P4(2) DDD_P4(3) DDD_P4(4) DDD_P4(5) DDD_P4(6) DDD_P4(7)
  DDD_P4(8) DDD_P4(9) DDD_P4(10) DDD_P4(11) DDD_P4(12) DDD_P4(13) DDD_P4(14) DDD_P4(15)
#undef DDD_P4
}

// What the relu of the MFMA path assumes of `v_pk_add_f32 x, 0 clamp` (rhs_mfma.h:
// activate16) and of the constant-lane-mask select (upper_half_one): out[0][i] / out[1][i]
// = the clamped pair built from in[2 i], in[2 i + 1]; out[2][lane] = the select of the lane id.
__global__ void relu_clamp_probe_kernel(const float* __restrict__ in, float* __restrict__ out) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int l = threadIdx.x;
  f32x2 v{in[2 * l], in[2 * l + 1]}, y;
  asm volatile("v_pk_add_f32 %0, %1, 0 clamp" : "=v"(y) : "v"(v));
  out[l] = y[0];
  out[64 + l] = y[1];
  float sel;
  const float x = (float)l;
  asm volatile("v_cndmask_b32_e64 %0, %1, 1.0, %2" : "=v"(sel) : "v"(x), "s"(0xffffffff00000000ull));
  out[128 + l] = sel;
}

}  // namespace ops
}  // namespace ddd
