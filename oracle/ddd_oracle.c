/*
 * CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE ONLY, never the product.
 *
 * Plain-C, float32 restatement of the reference's learned-stencil right-hand
 * side and fixed-step integrator, parallel over samples with OpenMP.  It is
 * what bench.py times as `cpu_baseline` (kind "port") and what
 * tests/test_cpu_oracle_c.py cross-checks against the NumPy oracle
 * (oracle/oracle.py), which in turn is pinned against the reference's golden
 * vectors.  Pinning status: same as oracle/oracle.py (see its header); this
 * file adds no independent authority.
 *
 * Follows (google/data-driven-discretization-1d, pde_superresolution/):
 *   conv tower          model.py:449-458, 492-495; layers.py:39-83, 103-137
 *   projection          polynomials.py:266-277 (bias + inputs @ nullspace)
 *   stencil apply       model.py:516-548 (patches, einsum 'bxdi,bxi->bxd')
 *   equation of motion  equations.py:269-274, 331-338, 410-415, 450-457,
 *                       518-524, 559-567 and the Godunov forms :341-370 etc.
 *   forcing             equations.py:214-219 evaluated on the reference grid
 *                       in float32 and resampled (mean / subsample)
 *   midpoint stepping   model.py:138-159 (tf.contrib odeint_fixed 'midpoint')
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  int equation, n, d, g, layers, filters, ksize, act, c_out;
  int fixed;           /* 1: `bias` holds fixed stencils [D][G], no conv net */
  int conservative, forced, nparams, resample_factor;
  float eta, stddev, inv_dx;
  double period;
  int in_start[4], in_size[4];
  const float* weights;   /* per layer: kernel [K][Cin][Cout], then bias [Cout] */
  const float* nullspace; /* per derivative [in_size][G], concatenated */
  const float* bias;      /* [D][G] */
} oracle_spec;

enum { ACT_RELU = 0, ACT_RELU6 = 1, ACT_TANH = 2, ACT_SOFTPLUS = 3, ACT_ELU = 4 };

static float activate(float x, int act) {
  switch (act) {
    case ACT_RELU: return x > 0.0f ? x : 0.0f;
    case ACT_RELU6: return x < 0.0f ? 0.0f : (x > 6.0f ? 6.0f : x);
    case ACT_TANH: return tanhf(x);
    case ACT_SOFTPLUS: return (x > 0.0f ? x : 0.0f) + log1pf(expf(-fabsf(x)));
    case ACT_ELU: return x > 0.0f ? x : expm1f(x);
    default: return x;
  }
}

static int wrap(int i, int n) {
  i %= n;
  return i < 0 ? i + n : i;
}

static float godunov(float um, float up) {
  const float a = um * um, b = up * up;
  return 0.5f * (um <= up ? fminf(a, b) : fmaxf(a, b));
}

/* u_t (plain forms) or the flux (flux forms) from the spatial derivatives. */
static float rhs_or_flux(int eq, float y, const float* d, float eta) {
  switch (eq) {
    case 0: return eta * d[1] - y * d[0];
    case 1: return 0.5f * (d[0] * d[0]) - eta * d[1];
    case 6: return godunov(d[0], d[1]) - eta * d[2];
    case 2: return (-6.0f * y) * d[0] - d[1];
    case 3: return 3.0f * (d[0] * d[0]) + d[1];
    case 7: return 6.0f * godunov(d[0], d[1]) + d[2];
    case 4: return (-y * d[0] - d[2]) - d[1];
    case 5: return (0.5f * (d[0] * d[0]) + d[2]) + d[1];
    case 8: return (d[3] + d[2]) + godunov(d[0], d[1]);
    default: return 0.0f;
  }
}

/* scratch per thread: two activation planes [n][cmax] + flux[n] + forcing[n*rf] */
static size_t scratch_floats(const oracle_spec* s) {
  int cmax = s->filters > s->c_out ? s->filters : s->c_out;
  if (cmax < 1) cmax = 1;
  return (size_t)2 * s->n * cmax + (size_t)s->n + (size_t)s->n * s->resample_factor;
}

/* One sample: out[n] = finalize(t, predict_time_derivative(y[n])). */
static void rhs_one(const oracle_spec* s, double t, const float* y,
                    const double* fa, const double* fomega, const double* fk,
                    const double* fphi, float* out, float* scratch) {
  const int n = s->n;
  int cmax = s->filters > s->c_out ? s->filters : s->c_out;
  if (cmax < 1) cmax = 1;
  float* cur = scratch;
  float* nxt = scratch + (size_t)n * cmax;
  float* flux = scratch + (size_t)2 * n * cmax;
  float* fref = flux + n;
  const float* net = NULL;

  if (!s->fixed) {
    for (int x = 0; x < n; ++x) cur[x] = y[x] / s->stddev;
    const float* w = s->weights;
    for (int l = 0; l < s->layers; ++l) {
      const int cin = l == 0 ? 1 : s->filters;
      const int cout = l == s->layers - 1 ? s->c_out : s->filters;
      const float* b = w + (size_t)s->ksize * cin * cout;
      const int act = l < s->layers - 1 ? s->act : -1;
      const int left = s->ksize / 2;
      for (int x = 0; x < n; ++x) {
        float* o = nxt + (size_t)x * cout;
        for (int co = 0; co < cout; ++co) o[co] = 0.0f;
        for (int k = 0; k < s->ksize; ++k) {
          const float* in = cur + (size_t)wrap(x + k - left, n) * cin;
          const float* wk = w + (size_t)k * cin * cout;
          for (int ci = 0; ci < cin; ++ci) {
            const float v = in[ci];
            const float* wr = wk + (size_t)ci * cout;
            for (int co = 0; co < cout; ++co) o[co] += v * wr[co];
          }
        }
        for (int co = 0; co < cout; ++co) o[co] = activate(o[co] + b[co], act);
      }
      w = b + cout;
      float* tmp = cur; cur = nxt; nxt = tmp;
    }
    net = cur;
  }

  const int gl = s->g / 2;
  for (int x = 0; x < n; ++x) {
    float dv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int d = 0; d < s->d; ++d) {
      float acc = 0.0f;
      for (int g = 0; g < s->g; ++g) {
        float coeff;
        if (s->fixed) {
          coeff = s->bias[d * s->g + g];
        } else {
          const float* ns = s->nullspace;
          for (int dd = 0; dd < d; ++dd) ns += (size_t)s->in_size[dd] * s->g;
          float proj = 0.0f;
          for (int j = 0; j < s->in_size[d]; ++j)
            proj += net[(size_t)x * s->c_out + s->in_start[d] + j] * ns[j * s->g + g];
          coeff = s->bias[d * s->g + g] + proj;
        }
        acc += coeff * y[wrap(x + g - gl, n)];
      }
      dv[d] = acc;
    }
    const float r = rhs_or_flux(s->equation, y[x], dv, s->eta);
    if (s->conservative) flux[x] = r; else out[x] = r;
  }
  if (s->conservative)
    for (int x = 0; x < n; ++x)
      out[x] = -(s->inv_dx * (flux[x + 1 == n ? 0 : x + 1] - flux[x]));

  if (s->forced && fa != NULL) {
    const int rf = s->resample_factor, nref = n * rf;
    const float tf = (float)t;
    for (int i = 0; i < nref; ++i) fref[i] = 0.0f;
    for (int m = 0; m < s->nparams; ++m) {
      const float a = (float)fa[m], ph = (float)fphi[m];
      const float wt = (float)fomega[m] * tf;
      /* spatial phase 2 pi k x / L in float64, cast to float32 like the TF graph */
      const double kscale = 2.0 * M_PI * fk[m] / nref;
      for (int i = 0; i < nref; ++i) {
        const float sp = (float)(kscale * i);
        fref[i] += a * sinf((wt + sp) + ph);
      }
    }
    for (int x = 0; x < n; ++x) {
      float f;
      if (s->conservative) {   /* Grid.resample 'mean' */
        float sum = 0.0f;
        for (int r = 0; r < rf; ++r) sum += fref[x * rf + r];
        f = sum / (float)rf;
      } else {                 /* 'subsample' */
        f = fref[x * rf];
      }
      out[x] += f;
    }
  }
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* forcing arrays [batch][nparams] in float64 (RandomForcing's dtype), or NULL */
void oracle_time_derivative(const oracle_spec* s, double t, const float* y,
                            const double* fa, const double* fomega,
                            const double* fk, const double* fphi, float* out,
                            int batch) {
#pragma omp parallel
  {
    float* scratch = (float*)malloc(scratch_floats(s) * sizeof(float));
#pragma omp for schedule(static)
    for (int b = 0; b < batch; ++b) {
      const size_t fo = (size_t)b * s->nparams;
      rhs_one(s, t, y + (size_t)b * s->n, fa ? fa + fo : NULL,
              fomega ? fomega + fo : NULL, fk ? fk + fo : NULL,
              fphi ? fphi + fo : NULL, out + (size_t)b * s->n, scratch);
    }
    free(scratch);
  }
}

/* Explicit RK in "previous stage only" form (same tableaus as oracle.py). */
static int tableau(int scheme, float* a, float* b, double* c) {
  switch (scheme) {
    case 0: a[0] = 0; b[0] = 1; c[0] = 0; return 1;
    case 1: a[0] = 0; a[1] = 0.5f; b[0] = 0; b[1] = 1; c[0] = 0; c[1] = 0.5; return 2;
    case 2:
      a[0] = 0; a[1] = 0.5f; a[2] = 0.75f;
      b[0] = (float)(2.0 / 9.0); b[1] = (float)(1.0 / 3.0); b[2] = (float)(4.0 / 9.0);
      c[0] = 0; c[1] = 0.5; c[2] = 0.75;
      return 3;
    case 3:
      a[0] = 0; a[1] = 0.5f; a[2] = 0.5f; a[3] = 1.0f;
      b[0] = (float)(1.0 / 6.0); b[1] = (float)(1.0 / 3.0);
      b[2] = (float)(1.0 / 3.0); b[3] = (float)(1.0 / 6.0);
      c[0] = 0; c[1] = 0.5; c[2] = 0.5; c[3] = 1.0;
      return 4;
    default: return 0;
  }
}

/* y [batch][n] is advanced in place by n_steps fixed steps (float32 state). */
int oracle_integrate_fixed(const oracle_spec* s, int scheme, double t0, double dt,
                           int n_steps, float* y, const double* fa,
                           const double* fomega, const double* fk,
                           const double* fphi, int batch) {
  float a[4], bb[4];
  double c[4];
  const int stages = tableau(scheme, a, bb, c);
  if (stages == 0) return -1;
  const float h = (float)dt;
  const int n = s->n;
#pragma omp parallel
  {
    float* scratch = (float*)malloc((scratch_floats(s) + 4 * (size_t)n) * sizeof(float));
    float* us = scratch + scratch_floats(s);
    float* f = us + n;
    float* ynew = f + n;
    float* kprev = ynew + n;
#pragma omp for schedule(static)
    for (int b = 0; b < batch; ++b) {
      float* yb = y + (size_t)b * n;
      const size_t fo = (size_t)b * s->nparams;
      for (int step = 0; step < n_steps; ++step) {
        const double t = t0 + (double)step * dt;
        memcpy(ynew, yb, n * sizeof(float));
        for (int st = 0; st < stages; ++st) {
          for (int x = 0; x < n; ++x)
            us[x] = st == 0 ? yb[x] : yb[x] + kprev[x] * (a[st] * h);
          rhs_one(s, t + c[st] * dt, us, fa ? fa + fo : NULL,
                  fomega ? fomega + fo : NULL, fk ? fk + fo : NULL,
                  fphi ? fphi + fo : NULL, f, scratch);
          if (bb[st] != 0.0f)
            for (int x = 0; x < n; ++x) ynew[x] = ynew[x] + (bb[st] * h) * f[x];
          memcpy(kprev, f, n * sizeof(float));
        }
        memcpy(yb, ynew, n * sizeof(float));
      }
    }
    free(scratch);
  }
  return 0;
}
