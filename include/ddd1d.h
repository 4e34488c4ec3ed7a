/*
 * ddd1d.h -- C ABI of libddd1d.so, the MI355X (gfx950) implementation of the
 * per-timestep learned-stencil integration path of
 * google/data-driven-discretization-1d (package pde_superresolution).
 *
 * The reference has no FFI: its plug-in seam for this path is the Python
 * callable `Differentiator.__call__(t, y) -> dy/dt` (integrate.py:40-45)
 * consumed by `integrate.odeint` (integrate.py:143-169), plus the batched
 * fixed-step `model.integrate_ode` (model.py:138-159).  Each entry point below
 * names the reference interface it replaces.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 *
 * Conventions
 *   - Every function returns 0 on success and a negative ddd_status on error;
 *     ddd_last_error() returns a thread-local human readable message.  No C++
 *     exception crosses this boundary.
 *   - `ddd_model*` is an opaque handle owned by the caller until
 *     ddd_model_destroy().  One handle per host thread / GPU; calls on one
 *     handle are stream ordered and must not be issued concurrently.
 *   - Unless marked HOST, pointers are device pointers (HBM) valid on the
 *     current HIP device.  `stream` is a hipStream_t passed as void* (NULL =
 *     the default stream).  Nothing here synchronises the device except
 *     ddd_model_create / ddd_set_forcing (which copy small host tables).
 *   - Batches are row major [batch][x] ("batch-major"): one sample's N grid
 *     points are contiguous, so a wavefront's 64 lanes read 256 contiguous
 *     bytes.  All data arithmetic is IEEE float32 (the reference's TF graph
 *     dtype); the `_f64` entry points keep the *integration state* in float64
 *     like SciPy does on the reference's host side.
 */
#ifndef DDD1D_H_
#define DDD1D_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDD_ABI_VERSION 1
/* libddd1d.so is built with -fvisibility=hidden: the entry points below are its
 * whole dynamic symbol table (tests/test_cpu_host_api.py compares `nm -D`
 * with this header). */
#if defined(__GNUC__)
#define DDD_API __attribute__((visibility("default")))
#else
#define DDD_API
#endif
#define DDD_MAX_DERIVATIVES 4 /* Godunov KS uses 4 (equations.py:570-574) */
#define DDD_MAX_STENCIL 16
#define DDD_MAX_LAYERS 8
#define DDD_MAX_STAGES 4

typedef enum ddd_status {
  DDD_OK = 0,
  DDD_ERR_INVALID_ARGUMENT = -1,
  DDD_ERR_UNSUPPORTED = -2,
  DDD_ERR_HIP = -3,
  DDD_ERR_NO_DEVICE = -4
} ddd_status;

/* equations.py: EQUATION_TYPES / CONSERVATIVE_EQUATION_TYPES / FLUX_EQUATION_TYPES
 * (:590-606).  Derivative order of the entries of `derivative_orders` must be
 * the class's DERIVATIVE_ORDERS. */
typedef enum ddd_equation {
  DDD_EQ_BURGERS = 0,              /* equations.py:230-302  (u_x, u_xx)        */
  DDD_EQ_BURGERS_CONSERVATIVE = 1, /* equations.py:323-338  (u, u_x)           */
  DDD_EQ_KDV = 2,                  /* equations.py:373-439  (u_x, u_xxx)       */
  DDD_EQ_KDV_CONSERVATIVE = 3,     /* equations.py:442-457  (u, u_xx)          */
  DDD_EQ_KS = 4,                   /* equations.py:481-548  (u_x, u_xx, u_xxxx)*/
  DDD_EQ_KS_CONSERVATIVE = 5,      /* equations.py:551-567  (u, u_x, u_xxx)    */
  DDD_EQ_BURGERS_GODUNOV = 6,      /* equations.py:352-370  (u-, u+, u_x)      */
  DDD_EQ_KDV_GODUNOV = 7,          /* equations.py:460-478  (u-, u+, u_xx)     */
  DDD_EQ_KS_GODUNOV = 8            /* equations.py:570-587  (u-, u+, u_x, u_xxx)*/
} ddd_equation;

/* model.py:411-417 (_NONLINEARITIES) */
typedef enum ddd_activation {
  DDD_ACT_RELU = 0,
  DDD_ACT_RELU6 = 1,
  DDD_ACT_TANH = 2,
  DDD_ACT_SOFTPLUS = 3,
  DDD_ACT_ELU = 4
} ddd_activation;

/* hparams.model_target, model.py:579-640 */
typedef enum ddd_model_target {
  DDD_TARGET_COEFFICIENTS = 0,
  DDD_TARGET_SPACE_DERIVATIVES = 1,
  DDD_TARGET_TIME_DERIVATIVE = 2,
  DDD_TARGET_FLUX = 3
} ddd_model_target;

/* Fixed-step explicit Runge-Kutta schemes.  MIDPOINT is
 * tf.contrib.integrate.odeint_fixed(method='midpoint') as called from
 * model.integrate_ode (model.py:155-157); BS3 is the Bogacki-Shampine tableau
 * of SciPy's RK23 (the reference's production integrator, integrate.py:154)
 * without step-size control. */
typedef enum ddd_scheme {
  DDD_SCHEME_EULER = 0,
  DDD_SCHEME_MIDPOINT = 1,
  DDD_SCHEME_BS3 = 2,
  DDD_SCHEME_RK4 = 3
} ddd_scheme;

typedef enum ddd_kernel_kind {
  DDD_KERNEL_AUTO = 0,    /* MFMA path when the configuration allows it */
  DDD_KERNEL_GENERIC = 1, /* any configuration, scalar FMA, one block / sample */
  DDD_KERNEL_MFMA = 2,    /* f32 MFMA tiles; 64 or 256 grid points / block    */
  DDD_KERNEL_MFMA_ROWS64 = 3,  /* force one-wavefront workgroups (N divides 64) */
  DDD_KERNEL_MFMA_ROWS256 = 4, /* force four-wavefront workgroups               */
  DDD_KERNEL_MFMA_ROWS64_W32 = 5, /* force 64-row groups on two 32-row wavefronts */
  DDD_KERNEL_MFMA_ROWS64_W16 = 6  /* force 64-row groups on FOUR 16-row wavefronts
                                     (small ensembles: one sample on the four SIMDs of
                                     a CU; per-equation float32 integrators only)      */
} ddd_kernel_kind;

/* How ddd_integrate_fixed advances time. */
typedef enum ddd_launch_mode {
  DDD_LAUNCH_PERSISTENT = 0, /* time loop inside ONE launch, state in registers */
  DDD_LAUNCH_PER_SUBSTEP = 1, /* one fused launch per RK substep, state in HBM  */
  DDD_LAUNCH_PER_STEP = 2 /* all stages of a step in one launch, state through HBM
                           * once per step (models on a per-equation MFMA kernel;
                           * others advance substep by substep)                */
} ddd_launch_mode;

/* Static description of one model: the equation (equations.py), the solution
 * grid (equations.py:44-68) and the conv-net hyper-parameters
 * (training.py:127-141). */
typedef struct ddd_config {
  int32_t struct_size; /* = sizeof(ddd_config), checked */
  int32_t equation;    /* ddd_equation */
  int32_t num_points;  /* N = grid.solution_num_points */
  int32_t num_derivatives;
  int32_t derivative_orders[DDD_MAX_DERIVATIVES];
  double dx;                 /* grid.solution_dx */
  double period;             /* grid.period */
  double eta;                /* Burgers viscosity, else 0 */
  double standard_deviation; /* Equation.standard_deviation (input scaling) */
  int32_t stencil_size;      /* G: 7 centered / 6 staggered by default */
  int32_t model_target;      /* ddd_model_target */
  int32_t num_layers;        /* conv layers incl. the linear output layer */
  int32_t filter_size;       /* hidden channels (32) */
  int32_t kernel_size;       /* conv taps (5) */
  int32_t activation;        /* ddd_activation */
  int32_t polynomial_accuracy_order;    /* 0: net emits D*G coefficients */
  int32_t ensure_unbiased_coefficients; /* only with accuracy order 0 */
  int32_t input_sizes[DDD_MAX_DERIVATIVES]; /* null-space dims per derivative */
  int32_t weno_reconstruction; /* ddd_baseline_create only: 1 = derivative slots
                                * 0 and 1 (u_minus, u_plus of the Godunov
                                * equations) are WENO5 reconstructions
                                * (weno.py:43-123, integrate.py:124-140,
                                * model.py:82-88) instead of stencil rows */
  int32_t reserved[3];
} ddd_config;

typedef struct ddd_model ddd_model;

/* ---- lifecycle ---------------------------------------------------------- */

/* Replaces: SavedModelDifferentiator.__init__ (integrate.py:51-68), i.e.
 * building model.predict_time_derivative (model.py:618-640) and restoring the
 * checkpoint.  HOST inputs:
 *   weights   layer-major; per layer the conv kernel [K][Cin][Cout] (the
 *             tf.layers.conv1d variable layout) followed by the bias [Cout].
 *             Cin = 1 for the first layer, filter_size otherwise; Cout =
 *             filter_size for hidden layers and, for the output layer,
 *             sum(input_sizes) (coefficients target, accuracy order > 0),
 *             D*G (accuracy order 0), D (space_derivatives) or 1.
 *   nullspace per derivative [input_size][G], float32 of
 *             PolynomialAccuracyLayer.nullspace (polynomials.py:246-264);
 *             NULL unless target = coefficients with accuracy order > 0.
 *   bias      [D][G] float32 of PolynomialAccuracyLayer.bias; same condition.
 * Which kernel family serves the model (ddd_kernel_name; training.py:127-141 leaves
 * these hyper-parameters free): 8 <= num_points <= 256, >= 2 layers, kernel_size <= 7 and
 * filter_size <= 64 (not both > 5 and > 32): the f32-MFMA kernels -- 5 taps x 32 filters with
 * per-equation kernels, 7 x 32 / 5 x 64 / 3 x 32 with streamed weights, nets in between
 * embedded exactly with zero weights; num_layers = 1 with coefficient output: affine
 * coefficients folded here, evaluated by the lane == grid point kernel (persistent
 * launches, num_points | 64, float32 state: "valu_f32_lean") or on the VALU route of the
 * MFMA-path kernels; everything else (larger nets or grids, num_points < 8, stencils > 12
 * points): the generic kernel.
 */
DDD_API int ddd_model_create(const ddd_config* cfg, const float* weights,
                     size_t n_weights, const float* nullspace,
                     size_t n_nullspace, const float* bias, size_t n_bias,
                     ddd_model** out);

/* Replaces: PolynomialDifferentiator.__init__ (integrate.py:77-103) /
 * model.baseline_space_derivatives (model.py:59-112): fixed stencils.
 * HOST `stencils` is [D][G] float32, each derivative's standard coefficients
 * (polynomials.coefficients) centred in a common G-wide window so that tap i
 * multiplies u[x + i - G/2] (the alignment of layers.pad_periodic(center=True),
 * layers.py:76-79).  Net fields of cfg are ignored.  Also used for
 * num_layers = 0 models (model.py:496-502) after folding on the host. */
DDD_API int ddd_baseline_create(const ddd_config* cfg, const float* stencils,
                        size_t n_stencils, ddd_model** out);

/* Replaces: integrate.SpectralDifferentiator.__init__ (integrate.py:110-111)
 * and the ExactMethod.SPECTRAL branch of model.baseline_space_derivatives
 * (model.py:78-80): the fine-grid "exact" solver of KdV / KS.  Float64.
 * HOST `kernels` is [D][N] float64: for each derivative the response of the
 * reference's spectral operator to a unit impulse at x = 0
 * (scipy.fftpack.diff(delta, order, period) for SpectralDifferentiator,
 * duckarray.spectral_derivative(delta, order, period) for the TF-graph form),
 * so that deriv[x] = sum_j kernels[d][(x - j) mod N] * y[j].
 * Non-flux equations only (DDD_EQ_BURGERS, DDD_EQ_KDV, DDD_EQ_KS:
 * integrate.py:346-347), N <= 2048.  Only the *_f64 entry points below accept
 * such a model; it carries no forcing (finalize_time_derivative stays on the
 * host, where the reference evaluates it in float64).
 * num_points a power of two >= 512: the same operators run as an in-LDS float64 FFT
 * (multipliers = the DFT of `kernels`, formed here in extended precision, so the
 * conventions of the call that produced the kernels carry over); smaller grids: the
 * O(N^2) circulant products. */
DDD_API int ddd_spectral_create(const ddd_config* cfg, const double* kernels,
                        size_t n_kernels, ddd_model** out);

DDD_API int ddd_model_destroy(ddd_model* model);

/* ---- per-sample forcing -------------------------------------------------
 * Replaces: RandomForcing.__call__ inside finalize_time_derivative
 * (equations.py:214-219, 276-277), one parameter row per sample.  HOST inputs,
 * all [batch][nparams]:
 *   amplitude  a_j            (times the exact block-mean factor when the grid
 *                              resamples by 'mean', i.e. conservative equations)
 *   omega      omega_j
 *   phase      phi_j          (plus the block-centre shift for 'mean')
 *   k_index    row of `spatial_phase` to use for mode j
 *   spatial_phase [n_k][N]    float32(2 pi k x_i / period) for each distinct k
 * forcing_b(x_i, t) = sum_j amplitude * sin((omega*t + spatial_phase) + phase),
 * accumulated in float32 in that order (the TF graph's order).  Ignored by
 * equations whose finalize_time_derivative is the identity (KdV, KS). */
DDD_API int ddd_set_forcing(ddd_model* model, int batch, int nparams,
                    const float* amplitude, const float* omega,
                    const float* phase, const int32_t* k_index,
                    const float* spatial_phase, int n_k);
DDD_API int ddd_clear_forcing(ddd_model* model);

/* ---- the hot path --------------------------------------------------------*/

/* Replaces: Differentiator.__call__(t, y) (integrate.py:70-71, 94-95), batched:
 * dydt[b] = finalize_time_derivative(t, predict_time_derivative(y[b])).
 * y, dydt: [batch][N] float32. */
DDD_API int ddd_time_derivative(ddd_model* model, double t, const float* y,
                        float* dydt, int batch, void* stream);

/* ONE fused launch = one Runge-Kutta substep:
 *     f       = time_derivative(t, y_in)
 *     y_out   = y_base + c1 * f          (y_base NULL -> c1 * f)
 *     acc_out = acc_in + c2 * f          (skipped when acc_out NULL;
 *                                         acc_in NULL -> c2 * f)
 * All arrays [batch][N] float32; y_out / acc_out may alias their inputs. */
DDD_API int ddd_rk_substep(ddd_model* model, double t, const float* y_in,
                   const float* y_base, float c1, float* y_out,
                   const float* acc_in, float c2, float* acc_out, int batch,
                   void* stream);

/* A caller that owns the Runge-Kutta loop -- the shape of integrate.odeint
 * (integrate.py:143-169: the driver calls the right-hand side once per stage) --
 * gets the library's own per-substep throughput by bracketing its loop:
 *
 *     ddd_stream_fork(model, stream);
 *     for (...) { ddd_rk_substep(model, ..., stream); ddd_rk_substep(model, ..., stream); }
 *     ddd_stream_join(model, stream);
 *
 * Inside the region, ddd_rk_substep / ddd_time_derivative calls on `stream`
 * advance a large ensemble as two contiguous half-ensembles on two internal
 * streams: half i of a call is ordered after everything enqueued on `stream`
 * before ddd_stream_fork and after half i of the previous call -- NOT after the
 * other half, so one half's launch boundary overlaps the other half's
 * evaluation.  Samples are independent, so results are bit-identical to the
 * unbracketed calls.  Contract inside the region: the arrays handed to these
 * calls are touched by nothing else (no other work on `stream` reads or writes
 * them, the host does not read them) until ddd_stream_join, after which `stream`
 * is ordered behind every substep enqueued.  Any other entry point taking this
 * model (and ddd_model_destroy) joins first, so a forgotten join cannot outlive
 * the model; calls on another stream, a batch too small to split, and models
 * without a per-equation MFMA kernel simply run on `stream` as without the
 * region.  ddd_stream_join without an open region is a no-op.  Neither call
 * synchronises the host.
 *
 * Round 6 -- the command ring (opt-in: ddd_set_region_mode(model, DDD_REGION_RING)).
 * Models with a per-equation MFMA kernel on one-wave groups (num_points divides 64;
 * default net) can run the region on ONE persistent kernel: the first ddd_rk_substep
 * of the region launches it on `stream`, and every call from then on is a 128-byte
 * command the host writes into a page-locked ring the kernel's wavefronts read -- no
 * launch, no drain, the conv weights stay in registers across calls.  Same contract as
 * above (the arrays belong to the region until the join), same bits as one launch per
 * call.  Measured at the rate of the launches (71.6 vs 71.1 % at 4 096 samples,
 * profiles/r6_ablation.txt), which is why it is not the default.  What differs:
 *   - work enqueued on `stream` INSIDE the region is ordered behind the persistent
 *     kernel, i.e. behind the join -- or behind the moment the region has been idle
 *     for 2 ms: a park thread then ends the kernel (the next call starts it again),
 *     so a caller that synchronises the device inside an open region waits that
 *     long, not for ever;
 *   - the host may run at most 256 calls ahead of the device; the 257th (and a join
 *     then) waits for room;
 *   - a host that stops for 20 s while the kernel waits (a debugger) trips the
 *     kernel's watchdog: the next call returns DDD_ERR_HIP and the region's results
 *     are undefined.
 * ddd_set_region_mode chooses: DDD_REGION_AUTO (default) and DDD_REGION_CHAINS = the
 * two chains of launches; DDD_REGION_RING = the ring where the model has one, else the
 * chains.  It closes an open region.  ddd_region_stats: persistent-kernel launches and
 * commands so far. */
enum ddd_region_mode { DDD_REGION_AUTO = 0, DDD_REGION_CHAINS = 1, DDD_REGION_RING = 2 };
DDD_API int ddd_stream_fork(ddd_model* model, void* stream);
DDD_API int ddd_stream_join(ddd_model* model, void* stream);
DDD_API int ddd_set_region_mode(ddd_model* model, int mode);
DDD_API int ddd_region_stats(const ddd_model* model, int64_t* ring_launches,
                             int64_t* ring_commands);

/* Replaces: model.integrate_ode (model.py:138-159) and, with the controller
 * pinned at max_step, the solve_ivp loop of integrate.odeint
 * (integrate.py:154-155) -- for the whole batch at once.
 * Advances n_steps steps of size dt from (t0, y0); every `save_every` steps
 * the state is written to y_out[(step+1)/save_every - 1][batch][N]
 * (n_steps / save_every snapshots).  y0 is not modified. */
DDD_API int ddd_integrate_fixed(ddd_model* model, int scheme, int launch_mode,
                        double t0, double dt, int n_steps, int save_every,
                        const float* y0, float* y_out, int batch, void* stream);

/* Same, integration state and I/O in float64 (right-hand side stays float32,
 * as in the reference where SciPy holds y in float64 and TF evaluates in
 * float32: integrate.py:57-60, 154).  Persistent launch mode only. */
DDD_API int ddd_integrate_fixed_f64(ddd_model* model, int scheme, double t0, double dt,
                            int n_steps, int save_every, const double* y0,
                            double* y_out, int batch, void* stream);

/* Replaces: integrate.odeint (integrate.py:143-169) =
 * scipy.integrate.solve_ivp(differentiator, (times[0], times[-1]), y0,
 * t_eval=times, max_step=0.01, method='RK23'), as scripts/run_evaluation.py
 * (:152-174) calls it once per sample -- for the whole batch in ONE launch,
 * with one SciPy-identical step-size controller per sample on the device:
 * Bogacki-Shampine 3(2) with FSAL, select_initial_step, error norm
 * RMS(err / (atol + rtol * max(|y|, |y_new|))), step factor
 * clip(0.9 * norm^(-1/3), 0.2, 10) without growth after a rejection, steps
 * clamped to [10 ulp(t), max_step], cubic dense output at `times`.  State,
 * controller and dense output in float64 (SciPy's), right-hand side in float32
 * fed float32(y), float32(t) (the TF placeholders, integrate.py:57-60).
 *   times  HOST [n_times], finite, strictly increasing; times[0] = t0 (the
 *          reference's rule: defaults rtol 1e-3, atol 1e-6, max_step 0.01)
 *   y0     [batch][N] float64;  y_out [n_times][batch][N] float64 (row 0 = y0)
 *   nfev   [batch] int32: right-hand-side evaluations of each sample
 *          (solve_ivp's sol.nfev)
 *   status [batch] int32: 0 = reached times[n_times-1]; -1 = step size fell
 *          below 10 ulp(t) (SciPy's "Required step size is less than spacing
 *          between numbers", sol.status -1); -2 = `max_attempts` steps tried
 *          (a safety net SciPy does not have; <= 0 selects a default of 1000x
 *          the attempts of a run at max_step).  Rows a failed sample did not
 *          reach are NaN, as integrate.odeint pads them (integrate.py:161-167).
 * Every model kind: MFMA-path models inside the persistent MFMA kernel; models
 * on the generic kernel (WENODifferentiator, integrate.py:124-140: the "exact"
 * Burgers solver; nets the MFMA path does not carry) with one workgroup per
 * sample; spectral models (ddd_spectral_create: SpectralDifferentiator, the
 * "exact" KdV / KS solver, integrate.py:108-121) with a float64 right-hand
 * side.  The spectral kernel carries no forcing term: a spectral model of the
 * Burgers family (whose finalize_time_derivative adds forcing(t),
 * equations.py:276-277) returns DDD_ERR_UNSUPPORTED here -- drive it from the host
 * over ddd_time_derivative_f64, as SpectralDifferentiator.__call__ does.
 * Enqueue-only: `times` is copied into a model-owned page-locked buffer before the
 * call returns (a ring of four; the host waits only if four adaptive calls on
 * this model are still in flight), uploaded and consumed on `stream`.  Not
 * capturable into a HIP graph (the ring recycles through events). */
DDD_API int ddd_integrate_adaptive_f64(ddd_model* model, const double* times,
                               int n_times, double rtol, double atol,
                               double max_step, long long max_attempts,
                               const double* y0, double* y_out, int32_t* nfev,
                               int32_t* status, int batch, void* stream);

/* Float64 forms for spectral models (ddd_spectral_create).
 * ddd_time_derivative_f64 replaces SpectralDifferentiator.__call__
 * (integrate.py:113-121) without finalize_time_derivative, batched;
 * ddd_rk_substep_f64 is ddd_rk_substep in float64:
 *   f = equation_of_motion(y_in);  y_out = y_base + c1 f;  acc_out = acc_in + c2 f
 * ddd_integrate_fixed_f64 on a spectral model steps with one fused launch per
 * substep, state and right-hand side in float64. */
DDD_API int ddd_time_derivative_f64(ddd_model* model, double t, const double* y,
                            double* dydt, int batch, void* stream);
DDD_API int ddd_rk_substep_f64(ddd_model* model, double t, const double* y_in,
                       const double* y_base, double c1, double* y_out,
                       const double* acc_in, double c2, double* acc_out,
                       int batch, void* stream);

/* Replaces: duckarray.smoothing_filter (duckarray.py:116-128), the low-pass
 * filter odeint_with_periodic_filtering (integrate.py:172-212) applies to the
 * state between segments and to the saved trajectory -- and any other
 * translation-invariant linear operator on the periodic grid.  Float64.
 *   out[r][x] = sum_j kernel[(x - j) mod n] * in[r][j],   r < rows
 * `kernel` [n] (device) is the operator applied to a unit impulse at x = 0
 * (smoothing_filter(delta, alpha, order)); in / out [rows][n]; out may alias in.
 * n <= 2048. */
DDD_API int ddd_circulant_apply_f64(const double* kernel, const double* in, double* out,
                            int rows, int n, void* stream);

/* ---- parity / debugging views of the same kernel ------------------------ */

/* Replaces: model.predict_space_derivatives (model.py:579-600) or
 * baseline_space_derivatives; out [batch][N][D]. */
DDD_API int ddd_space_derivatives(ddd_model* model, const float* y, float* out,
                          int batch, void* stream);

/* Replaces: model.predict_coefficients (model.py:420-513);
 * out [batch][N][D][G]. */
DDD_API int ddd_coefficients(ddd_model* model, const float* y, float* out, int batch,
                     void* stream);

/* ---- standalone operators (reference unit-test surface) ------------------ */

/* Replaces: layers.nn_conv1d_periodic / conv1d_periodic_layer
 * (layers.py:95-137).  in [batch][N][Cin], filters [K][Cin][Cout] (device),
 * bias [Cout] or NULL, out [batch][N][Cout]; activation -1 = none. */
DDD_API int ddd_conv1d_periodic(const float* in, const float* filters,
                        const float* bias, float* out, int batch, int n,
                        int cin, int cout, int k, int center, int activation,
                        void* stream);

/* Replaces: layers.pad_periodic (layers.py:39-83).
 * in [batch][N][C] -> out [batch][N + padding][C]. */
DDD_API int ddd_pad_periodic(const float* in, float* out, int batch, int n, int c,
                     int padding, int center, void* stream);

/* Replaces: model.extract_patches (model.py:516-533).
 * in [batch][N] -> out [batch][N][size], out[b][x][i] = in[b][(x + i - size/2) mod N]. */
DDD_API int ddd_extract_patches(const float* in, float* out, int batch, int n, int size,
                        void* stream);

/* Replaces: model.apply_coefficients (model.py:536-548).
 * coefficients [batch][N][D][G], in [batch][N] -> out [batch][N][D]. */
DDD_API int ddd_apply_coefficients(const float* coefficients, const float* in, float* out,
                           int batch, int n, int d, int g, void* stream);

/* Replaces: model.apply_space_derivatives (model.py:115-135):
 * Equation.equation_of_motion on given derivatives [batch][N][D] (order of
 * DERIVATIVE_NAMES) and the state y [batch][N] -> time derivative [batch][N],
 * without finalize_time_derivative.  `equation` is a ddd_equation. */
DDD_API int ddd_apply_space_derivatives(int equation, const float* derivatives, const float* y,
                                float* out, int batch, int n, int d, double eta,
                                double dx, void* stream);

/* Replaces: PolynomialAccuracyLayer.apply (polynomials.py:266-277).
 * inputs [m][input_size], nullspace [input_size][G], bias [G], out [m][G]. */
DDD_API int ddd_polynomial_accuracy_apply(const float* inputs, const float* nullspace,
                                  const float* bias, float* out, int64_t m,
                                  int input_size, int g, void* stream);

/* ---- introspection -------------------------------------------------------*/
DDD_API int ddd_set_kernel(ddd_model* model, int kernel_kind);
/* "mfma_f32_r64", "mfma_f32_r64w32", "mfma_f32_r64w16" (small ensembles: every 64-row group
 * on two / four wavefronts), "mfma_f32_r64h16" (nets of up to 16 filters on the block-diagonal
 * tower, persistent integrators), "mfma_f32_r256", "generic", "valu_f32_lean" (fixed
 * stencils / one-layer nets, persistent launch: lane == grid point, no matrix work),
 * "valu_f32_weno" (WENO5 + Godunov flux, integrate.py:124-140: one wavefront per sample),
 * "stream_fixed" (fixed stencils, one launch per substep or step) or "spectral_f64":
 * the kernel family and workgroup geometry of the most recent launch on this
 * handle (the automatic choice depends on the batch size and launch mode). */
DDD_API const char* ddd_kernel_name(const ddd_model* model);
/* Algorithmic multiply-adds per grid point per right-hand-side evaluation
 * (SURVEY.md section 8(d)); 2x this is the FLOP count used for the roofline. */
DDD_API int64_t ddd_fma_per_point(const ddd_model* model);
/* Number of right-hand-side evaluations per step of a scheme. */
DDD_API int ddd_scheme_stages(int scheme);
/* Runs tiny MFMA probes on the current device and checks the operand/result
 * register layouts the kernels assume; 0 = layouts as assumed. */
DDD_API int ddd_selftest_mfma_layout(void);
DDD_API int ddd_abi_version(void);
DDD_API const char* ddd_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DDD1D_H_ */
