// Spectral "exact" right-hand side in float64:
//   integrate.SpectralDifferentiator   integrate.py:108-121
//   duckarray.spectral_derivative      duckarray.py:105-113 (model.py:78-80)
// Both reference forms are linear, translation-invariant operators on a
// periodic grid, i.e. circulant matrices: deriv_d[x] = sum_j c_d[(x - j) mod N] y[j]
// with c_d = the operator applied to a unit impulse (computed on the host with
// the very SciPy / NumPy call the reference uses, so Nyquist conventions carry
// over).  One workgroup per sample; y and the D kernels sit in LDS as float64;
// each thread owns grid points and runs D dot products of length N (N <= 2048).
// The reference evaluates this path in float64 NumPy; so does this kernel.
#pragma once
#include "dev_params.h"

namespace ddd {
namespace spectral {

constexpr int kThreads = 256;
constexpr int kMaxPoints = 2048;

struct Params {
  int equation, N, D;
  double eta;
  const double* kernels;   // [D][N] circulant first columns
};

struct SubstepArgs64 {
  const double* y_in;
  const double* y_base;   // may be null
  double c1;
  double* y_out;          // may be null
  const double* acc_in;   // may be null
  double c2;
  double* acc_out;        // may be null
  int batch;
};

__host__ __device__ inline size_t lds_bytes(const Params& p) {
  return (size_t)(1 + p.D) * p.N * sizeof(double);
}

// equation_of_motion of the non-flux forms (equations.py:269-274, 410-415,
// 519-526) in the reference's expression order.
__device__ __forceinline__ double equation_rhs(int eq, double y, const double (&d)[kMaxDerivs],
                                               double eta) {
  switch (eq) {
    case EQ_BURGERS: return eta * d[1] - y * d[0];
    case EQ_KDV: return (-6.0 * y) * d[0] - d[1];
    default: return (-y * d[0] - d[2]) - d[1];   // EQ_KS: -u u_x - u_xxxx - u_xx
  }
}

__global__ __launch_bounds__(kThreads) void substep_kernel(Params p, SubstepArgs64 a) {
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  double* y = smem64;
  double* c = smem64 + p.N;
  const int n = p.N;
  const size_t off = (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += kThreads) y[i] = a.y_in[off + i];
  for (int i = threadIdx.x; i < p.D * n; i += kThreads) c[i] = p.kernels[i];
  __syncthreads();
  for (int pos = threadIdx.x; pos < n; pos += kThreads) {
    double dv[kMaxDerivs] = {0.0, 0.0, 0.0, 0.0};
    int idx = pos;   // (pos - j) mod N
    for (int j = 0; j < n; ++j) {
      const double yj = y[j];
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d)
        if (d < p.D) dv[d] = fma(c[d * n + idx], yj, dv[d]);
      idx = idx == 0 ? n - 1 : idx - 1;
    }
    const double f = equation_rhs(p.equation, y[pos], dv, p.eta);
    const size_t gi = off + pos;
    if (a.y_out != nullptr) {
      const double cf = a.c1 * f;
      a.y_out[gi] = a.y_base != nullptr ? a.y_base[gi] + cf : cf;
    }
    if (a.acc_out != nullptr) {
      const double cf = a.c2 * f;
      a.acc_out[gi] = a.acc_in != nullptr ? a.acc_in[gi] + cf : cf;
    }
  }
}

}  // namespace spectral
}  // namespace ddd
