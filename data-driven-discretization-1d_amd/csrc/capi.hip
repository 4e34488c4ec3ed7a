// C ABI of libddd1d.so (include/ddd1d.h): model lifecycle, weight packing for
// the MFMA kernels, launch dispatch.  No torch types, no exceptions across the
// boundary.
#include <hip/hip_runtime.h>

#include <emmintrin.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/ddd1d.h"
#include "dev_params.h"
#include "launch.h"
#include "launch_weno.h"
#include "ops.h"
#ifdef DDD_PROBES
#include "probe_kernels.h"
#endif
#include "rhs_generic.h"
#include "rhs_lean.h"
#include "rhs_mfma.h"
#include "rhs_spectral.h"
#include "rhs_stream.h"
#include "ring_args.h"
#include "rhs_weno.h"   // (host side only: weno::supports; the kernels are weno_unit.hip)

namespace {

thread_local std::string g_error;

// Profiling / A-B switches.  The product library (libddd1d.so) has none: the
// struct below is constant there and every test on it folds away.  In
// libddd1d_probe.so (-DDDD_PROBES, __graft_entry__.build_probe, used only by
// profiles/tools/ and `bench.py --debug-option`) they are set ONLY through
// ddd_debug_set_option (an explicit call that logs to stderr), never read from
// the environment: a stray variable must not be able to change the numerics
// path, skip kernel phases or hand the kernel a raw address.
struct DebugOptions {
  int no_fold = 0;       // keep the projection out of the output layer (D <= 2 models)
  int no_spec = 0;       // run-time-parameterised kernels instead of the per-equation ones
  int no_stream = 0;     // per-sample kernels instead of the streaming fixed-stencil kernel
  int no_lean = 0;       // the MFMA-path kernels (tower skipped) instead of rhs_lean.h
  int no_weno = 0;       // the generic kernel instead of rhs_weno.h
  int no_fft = 0;        // spectral models: the O(N^2) circulant form at every N
  int no_half = 0;       // nets of <= 16 filters: embedded in 32 instead of the block-diagonal tower
  int prio_split = 0;    // A/B: static wave priorities
  int stagger = 0;       // A/B: initial s_sleep of odd wave slots
  int substep_parts = 0; // A/B: sample slabs advanced side by side in the per-substep modes (0: auto)
  int ablate = 0;        // skips kernel phases: WRONG RESULTS (run-time-parameterised kernels)
  unsigned long long trace_ptr = 0;   // device buffer for s_memtime phase stamps
  unsigned long long walk_trace_ptr = 0;   // device buffer for the wave-lifetime stamps of the substep walk
};
#ifdef DDD_PROBES
DebugOptions g_debug;
#else
constexpr DebugOptions g_debug{};
#endif

int fail(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_error = buf;
  return code;
}

#define DDD_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t err__ = (expr);                                               \
    if (err__ != hipSuccess)                                                 \
      return fail(DDD_ERR_HIP, "%s failed: %s (%s:%d)", #expr,              \
                  hipGetErrorString(err__), __FILE__, __LINE__);             \
  } while (0)

template <typename T>
int upload(const std::vector<T>& host, T** dev) {
  *dev = nullptr;
  if (host.empty()) return DDD_OK;
  DDD_HIP(hipMalloc(reinterpret_cast<void**>(dev), host.size() * sizeof(T)));
  DDD_HIP(hipMemcpy(*dev, host.data(), host.size() * sizeof(T),
                    hipMemcpyHostToDevice));
  return DDD_OK;
}

bool is_conservative(int eq) {
  switch (eq) {
    case DDD_EQ_BURGERS_CONSERVATIVE: case DDD_EQ_KDV_CONSERVATIVE:
    case DDD_EQ_KS_CONSERVATIVE: case DDD_EQ_BURGERS_GODUNOV:
    case DDD_EQ_KDV_GODUNOV: case DDD_EQ_KS_GODUNOV:
      return true;
    default:
      return false;
  }
}

bool is_forced_family(int eq) {   // equations.py:276-277: only Burgers adds forcing(t)
  return eq == DDD_EQ_BURGERS || eq == DDD_EQ_BURGERS_CONSERVATIVE ||
         eq == DDD_EQ_BURGERS_GODUNOV;
}

int expected_derivatives(int eq) {
  switch (eq) {
    case DDD_EQ_BURGERS: case DDD_EQ_BURGERS_CONSERVATIVE: case DDD_EQ_KDV:
    case DDD_EQ_KDV_CONSERVATIVE:
      return 2;
    case DDD_EQ_KS: case DDD_EQ_KS_CONSERVATIVE: case DDD_EQ_BURGERS_GODUNOV:
    case DDD_EQ_KDV_GODUNOV:
      return 3;
    case DDD_EQ_KS_GODUNOV:
      return 4;
    default:
      return -1;
  }
}

// b weights in float64 (the float32 tableau rounds 2/9, 1/3, 4/9, 1/6).
void tableau_weights_f64(int scheme, double (&b)[ddd::kMaxStages]) {
  for (double& v : b) v = 0.0;
  switch (scheme) {
    case DDD_SCHEME_EULER: b[0] = 1.0; break;
    case DDD_SCHEME_MIDPOINT: b[1] = 1.0; break;
    case DDD_SCHEME_BS3: b[0] = 2.0 / 9.0; b[1] = 1.0 / 3.0; b[2] = 4.0 / 9.0; break;
    case DDD_SCHEME_RK4: b[0] = 1.0 / 6.0; b[1] = 1.0 / 3.0; b[2] = 1.0 / 3.0; b[3] = 1.0 / 6.0; break;
    default: break;
  }
}

int make_tableau(int scheme, ddd::Tableau* tab) {
  std::memset(tab, 0, sizeof(*tab));
  switch (scheme) {
    case DDD_SCHEME_EULER:
      tab->stages = 1; tab->b[0] = 1.0f;
      return DDD_OK;
    case DDD_SCHEME_MIDPOINT:   // tf.contrib.integrate.odeint_fixed 'midpoint'
      tab->stages = 2;
      tab->a[1] = 0.5f; tab->c[1] = 0.5;
      tab->b[0] = 0.0f; tab->b[1] = 1.0f;
      return DDD_OK;
    case DDD_SCHEME_BS3:        // Bogacki-Shampine (SciPy RK23 tableau)
      tab->stages = 3;
      tab->a[1] = 0.5f; tab->a[2] = 0.75f;
      tab->c[1] = 0.5; tab->c[2] = 0.75;
      tab->b[0] = (float)(2.0 / 9.0); tab->b[1] = (float)(1.0 / 3.0);
      tab->b[2] = (float)(4.0 / 9.0);
      return DDD_OK;
    case DDD_SCHEME_RK4:
      tab->stages = 4;
      tab->a[1] = 0.5f; tab->a[2] = 0.5f; tab->a[3] = 1.0f;
      tab->c[1] = 0.5; tab->c[2] = 0.5; tab->c[3] = 1.0;
      tab->b[0] = (float)(1.0 / 6.0); tab->b[1] = (float)(1.0 / 3.0);
      tab->b[2] = (float)(1.0 / 3.0); tab->b[3] = (float)(1.0 / 6.0);
      return DDD_OK;
    default:
      return fail(DDD_ERR_INVALID_ARGUMENT, "unknown scheme %d", scheme);
  }
}

// The per-stage products of a fixed-step launch in the kernels' own arithmetic (float
// state: a[s] * (float)dt in float32; float64 state: (double)a[s] * dt; stage times
// c[s] * dt in float64) -- dev_params.h: StageConsts.
ddd::StageConsts make_stage_consts(const ddd::Tableau& tab, double dt) {
  ddd::StageConsts sc{};
  const float h = (float)dt;
  for (int s = 0; s < ddd::kMaxStages; ++s) {
    const volatile float ah = tab.a[s] * h, bh = tab.b[s] * h;   // (volatile: one rounding each,
    const volatile double ahd = (double)tab.a[s] * dt;            //  no host-side contraction)
    const volatile double bhd = (double)tab.b[s] * dt, ct = tab.c[s] * dt;
    sc.ah[s] = ah; sc.bh[s] = bh; sc.ahd[s] = ahd; sc.bhd[s] = bhd; sc.ct[s] = ct;
    if (s < tab.stages && tab.b[s] != 0.0f) sc.b_nonzero |= 1 << s;
  }
  return sc;
}

// Host side of the device-resident command ring (dev_params.h: RingArgs; rhs_mfma.h:
// substep_ring_kernel).  The caller's thread posts commands; a park thread (one per model that
// ever used the ring) posts kRingStop when the region has been idle for `idle` -- a caller
// that synchronises the device, or enqueues other work behind the persistent kernel, inside
// an open region waits that long, not for ever; the next ddd_rk_substep starts the kernel again.
struct Ring {
  void* slots = nullptr;         // page-locked, device-visible: [kRingSlots][kRingSlotChunks] x 16 B
  unsigned* done = nullptr;      // page-locked: [kRingSlots], written by the kernel
  unsigned* status = nullptr;    // page-locked: [0] watchdog expired
  unsigned* d_count = nullptr;   // device: [kRingSlots] + the relay lock
  void* d_slots = nullptr;       // device (fine-grained): the relayed commands
  std::mutex mu;
  std::condition_variable cv;
  std::thread parker;
  bool quit = false;
  bool running = false;          // a substep_ring_kernel is waiting for commands on `stream`
  bool dead = false;             // the watchdog expired: results of the region are undefined
  hipStream_t stream = nullptr;
  unsigned long long next_index = 0;    // index of the next command
  unsigned long long launch_index = 0;  // first command of the running kernel
  int grid = 0;
  std::chrono::steady_clock::time_point last_post;
  std::chrono::microseconds idle{2000};
  unsigned watchdog_ms = 20000;
  unsigned long long launches = 0, commands = 0;   // statistics (ddd_region_stats)
};

// One 16-byte store per chunk (see dev_params.h): payload dwords 3 c .. 3 c + 2, then the tag.
void ring_write(Ring* rg, unsigned long long index, const unsigned (&payload)[3 * ddd::kRingChunks]) {
  const unsigned tag = (unsigned)(index + 1ull);
  __m128i* slot = reinterpret_cast<__m128i*>(rg->slots) +
                  (size_t)(index & (ddd::kRingSlots - 1)) * ddd::kRingSlotChunks;
  for (int c = 0; c < ddd::kRingChunks; ++c)
    _mm_store_si128(slot + c, _mm_set_epi32((int)tag, (int)payload[3 * c + 2],
                                            (int)payload[3 * c + 1], (int)payload[3 * c]));
}

// Room for command `index`: the command kRingSlots before it has been passed by every group
// (or belongs to a kernel that has ended).  False: no progress for 10 s.
bool ring_wait_room(Ring* rg, unsigned long long index) {
  if (index < (unsigned long long)ddd::kRingSlots) return true;
  const unsigned long long old = index - ddd::kRingSlots;
  // (no shortcut for commands of an earlier launch: that kernel may still be working through
  // them while the next one is already enqueued behind it)
  const volatile unsigned* flag = rg->done + (old & (ddd::kRingSlots - 1));
  const unsigned want = (unsigned)(old + 1ull);
  if (*flag == want) return true;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned spin = 0; *flag != want; ++spin) {
    _mm_pause();
    if ((spin & 0xfff) == 0xfff) {
      if (*reinterpret_cast<const volatile unsigned*>(rg->status) != 0u) return false;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(10)) return false;
    }
  }
  return true;
}

// (rg->mu held)  Ends the running kernel: kRingStop is a command like any other, every group
// leaves after the commands before it.
bool ring_stop_locked(Ring* rg) {
  if (!rg->running) return true;
  rg->running = false;
  const bool room = ring_wait_room(rg, rg->next_index);
  unsigned payload[3 * ddd::kRingChunks] = {};
  payload[14] = (unsigned)ddd::kRingStop;
  // (without room the slot is overwritten all the same: the kernel must end, and a region
  // that made no progress for 10 s is reported as failed)
  ring_write(rg, rg->next_index, payload);
  ++rg->next_index;
  if (!room) rg->dead = true;
  return room;
}

void ring_parker(Ring* rg) {
  std::unique_lock<std::mutex> lk(rg->mu);
  while (!rg->quit) {
    if (!rg->running) { rg->cv.wait(lk); continue; }
    const auto deadline = rg->last_post + rg->idle;
    if (std::chrono::steady_clock::now() >= deadline) { (void)ring_stop_locked(rg); continue; }
    rg->cv.wait_until(lk, deadline);
  }
}

}  // namespace

struct ddd_model {
  ddd_config cfg;
  ddd::DevParams dp;
  bool mfma_ok = false;
  std::string mfma_reason;
  int kernel = DDD_KERNEL_GENERIC;   // resolved family
  int force_rows = 0;                // 0 = automatic; 64 / 32 (64 rows on two waves) / 256
  bool explicit_kernel = false;      // ddd_set_kernel chose a family (disables automatic variants)
  bool spec_folded = false;          // w_final4 (specialised kernels) holds the folded output layer
  bool last_launch_streamed = false; // the most recent launch was the streaming fixed-stencil kernel
  bool last_launch_split = false;    // ... the persistent integrator with two 32-row wavefronts per sample
  bool last_launch_lean = false;     // ... the lane == grid point kernel of rhs_lean.h
  bool last_launch_quad = false;     // ... four 16-row wavefronts per group (rhs_mfma.h kQuad)
  int last_batch = 0;                // batch of the most recent launch (kernel_name)
  int64_t fma_per_point = 0;
  // device allocations
  float* d_weights = nullptr;
  float* d_weights4 = nullptr;      // channel-padded copy for the generic kernel's LDS staging
  float* d_nullspace = nullptr;
  float* d_bias = nullptr;
  float* d_w_input = nullptr;
  float* d_w_hidden = nullptr;
  float* d_w_final4 = nullptr;
  float* d_w_final4_rt = nullptr;
  float* d_w_final4_split = nullptr;
  float* d_w_quad = nullptr;
  float* d_w_hidden_half = nullptr;   // block-diagonal packing of a <= 16-filter net (rhs_mfma.h HalfTower)
  float* d_w_final4_half = nullptr;
  float* d_w_t16 = nullptr;           // the same net as 16x16x4 A operands (rhs_mfma.h Tile16Tower)
  bool last_launch_half = false;
  bool wide = false;                 // run-time kernels of the wide flavour (rhs_mfma.h kWide)
  bool split_auto = true;            // small ensembles: two 32-row wavefronts per sample (kSplit)
  int tower_k = 5, tower_cb = 1;     // conv tower the MFMA kernels carry the net in (rhs_mfma.h Tower)
  bool big() const { return tower_k != ddd::mfma::kKW || tower_cb != 1; }
  float4* d_frc = nullptr;
  float* d_sp = nullptr;
  float* d_trig = nullptr;
  unsigned char* d_runs = nullptr;
  // spectral (float64) models: ddd_spectral_create
  bool spectral = false;
  ddd::spectral::Params sp{};
  double* d_kernels = nullptr;
  double2* d_fft_mult = nullptr;      // FFT mode (rhs_spectral.h): multipliers [D][N] and
  double2* d_fft_twiddle = nullptr;   // the twiddle table [N / 2]
  double* d_scratch64 = nullptr;
  size_t scratch64_doubles = 0;
  // per-substep launch mode: the ensemble is advanced as two half-ensembles on two
  // internal streams, so that one half's launch boundary overlaps the other's
  // steady state (ddd_integrate_fixed)
  hipStream_t aux_stream[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  // ddd_stream_fork .. ddd_stream_join: the same two chains kept alive ACROSS
  // ddd_rk_substep / ddd_time_derivative calls of a caller that owns the RK loop
  struct Chain {
    bool open = false;        // between ddd_stream_fork and ddd_stream_join
    bool forked = false;      // the internal streams are running ahead of `stream`
    hipStream_t stream = nullptr;
    int batch = 0, halves = 1;
    int half_batch[4] = {0, 0, 0, 0}, slab_first[4] = {0, 0, 0, 0};
  } chain;
  // ... or, for the per-equation one-wave kernels, ONE persistent kernel fed through a
  // device-resident command ring (Ring above)
  Ring* ring = nullptr;
  int region_mode = DDD_REGION_AUTO;
  // output times of ddd_integrate_adaptive_f64: a small ring of (page-locked host
  // copy, device copy) pairs, each released by an event recorded behind the launch
  // that reads it -- the entry point only enqueues (no host synchronisation unless
  // kTimeSlots calls are still in flight), on whatever stream each call names
  static constexpr int kTimeSlots = 4;
  struct TimeSlot {
    double* host = nullptr;    // hipHostMalloc
    double* dev = nullptr;
    size_t capacity = 0;
    hipEvent_t done = nullptr;
    bool in_flight = false;
  } time_slot[kTimeSlots];
  int next_time_slot = 0;
  // scratch for the per-substep launch mode
  float* d_scratch = nullptr;
  size_t scratch_floats = 0;
};

namespace {

void free_dev(void* p) { if (p) (void)hipFree(p); }

int common_config_checks(const ddd_config* cfg) {
  if (cfg == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "cfg is NULL");
  if (cfg->struct_size != (int32_t)sizeof(ddd_config))
    return fail(DDD_ERR_INVALID_ARGUMENT,
                "ddd_config.struct_size = %d, library expects %d (ABI mismatch)",
                cfg->struct_size, (int)sizeof(ddd_config));
  const int nd = expected_derivatives(cfg->equation);
  if (nd < 0) return fail(DDD_ERR_INVALID_ARGUMENT, "unknown equation %d", cfg->equation);
  if (cfg->num_derivatives != nd)
    return fail(DDD_ERR_INVALID_ARGUMENT,
                "equation %d has %d spatial derivatives, got num_derivatives = %d",
                cfg->equation, nd, cfg->num_derivatives);
  if (cfg->num_points < 2)
    return fail(DDD_ERR_INVALID_ARGUMENT, "num_points = %d", cfg->num_points);
  if (!(cfg->dx > 0.0))
    return fail(DDD_ERR_INVALID_ARGUMENT, "dx must be positive");
  if (cfg->stencil_size < 1 || cfg->stencil_size > DDD_MAX_STENCIL)
    return fail(DDD_ERR_INVALID_ARGUMENT, "stencil_size = %d out of range [1, %d]",
                cfg->stencil_size, DDD_MAX_STENCIL);
  return DDD_OK;
}

// rhs_mfma.h replaces u / stddev by q' = fma(fma(-q, s, u), r, q), q = RN(u r),
// r = RN(1 / s).  True for every float32 u iff it is true for the 2^23
// significands of one binade (scaling u by a power of two scales q, the
// residual and q' exactly, away from under/overflow).  ~0.1 s, cached per value.
bool division_shortcut_is_exact(float s) {
  static std::vector<std::pair<float, bool>> cache;
  for (const auto& kv : cache)
    if (std::memcmp(&kv.first, &s, sizeof(s)) == 0) return kv.second;
  const float r = (float)(1.0 / (double)s);
  bool exact = std::isfinite(r) && r != 0.0f;
  for (uint32_t bits = 0x3f800000u; exact && bits < 0x40000000u; ++bits) {
    float u;
    std::memcpy(&u, &bits, sizeof(u));
    const float q = u * r;
    const float fast = std::fmaf(std::fmaf(-q, s, u), r, q);
    exact = fast == u / s;
  }
  cache.emplace_back(s, exact);
  return exact;
}

void fill_equation(const ddd_config& cfg, ddd::DevParams* dp) {
  std::memset(dp, 0, sizeof(*dp));
  dp->equation = cfg.equation;
  dp->N = cfg.num_points;
  dp->D = cfg.num_derivatives;
  dp->G = cfg.stencil_size;
  dp->eta = (float)cfg.eta;
  dp->stddev = (float)cfg.standard_deviation;
  dp->inv_stddev = (float)(1.0 / (double)dp->stddev);
  dp->exact_div = division_shortcut_is_exact(dp->stddev) ? 0 : 1;
  dp->inv_dx = (float)(1.0 / cfg.dx);
  dp->conservative = is_conservative(cfg.equation) ? 1 : 0;
}

// Projection tables zero-padded to the kernels' stencil columns for the
// MFMA-path epilogue; they travel inside DevParams (kernel-argument segment).
// `identity`: polynomial_accuracy_order = 0 on the wide kernels -- output channel
// G d + g IS coefficient g of derivative d (model.py:460-475), expressed as a
// projection with unit rows and zero bias so that the unfolded epilogue applies.
int upload_padded_tables(ddd_model* m, const float* nullspace, const float* bias,
                         bool identity = false) {
  ddd::DevParams& dp = m->dp;
  std::memset(dp.ns8, 0, sizeof(dp.ns8));
  std::memset(dp.bias8, 0, sizeof(dp.bias8));
  dp.dsel_bits = 0;
  dp.dsel_valid = 0;
  if (bias != nullptr)
    for (int d = 0; d < dp.D; ++d)
      for (int g = 0; g < dp.G; ++g) dp.bias8[d][g] = bias[d * dp.G + g];
  if (identity) {
    for (int d = 0; d < dp.D; ++d)
      for (int g = 0; g < dp.G; ++g) {
        const int c = d * dp.G + g;
        dp.dsel_bits |= (unsigned long long)d << (2 * c);
        dp.dsel_valid |= 1u << c;
        dp.ns8[c][g] = 1.0f;
      }
  } else if (nullspace != nullptr) {
    for (int d = 0; d < dp.D; ++d)
      for (int j = 0; j < dp.in_size[d]; ++j) {
        const int c = dp.in_start[d] + j;
        dp.dsel_bits |= (unsigned long long)d << (2 * c);
        dp.dsel_valid |= 1u << c;
        for (int g = 0; g < dp.G; ++g)
          dp.ns8[c][g] = nullspace[dp.ns_off[d] + j * dp.G + g];
      }
  }
  return DDD_OK;
}

// [rows][64] -> the storage order of rhs_mfma.h load_rows4: four rows to a float4 per
// lane, rows zero-padded to a multiple of four.
std::vector<float> quad_rows(const float* rows64, int rows) {
  std::vector<float> out((size_t)ddd::mfma::padded_rows4(rows) * 64, 0.0f);
  for (int s = 0; s < rows; ++s)
    for (int lane = 0; lane < 64; ++lane)
      out[((size_t)(s >> 2) * 64 + lane) * 4 + (s & 3)] = rows64[(size_t)s * 64 + lane];
  return out;
}

// Natural-layout description of the net the MFMA packing reads: the model's own
// ([K][cin][cout] + bias per layer, DevParams::w_off / b_off) or its zero-padded
// 5-tap x 32-channel embedding (embed_small_tower).
struct NetLayout {
  const float* weights;
  int w_off[ddd::kMaxLayers], b_off[ddd::kMaxLayers];
};

// Reorder conv weights into MFMA A-operand order (rhs_mfma.h).
int pack_mfma_weights(ddd_model* m, const NetLayout& net) {
  const ddd::DevParams& dp = m->dp;
  const float* weights = net.weights;
  const int hidden = dp.L - 2;
  const int tk = m->tower_k, tcb = m->tower_cb, tc = 32 * tcb;   // the (padded) net: tk taps, tc filters
  // relu towers run on activations scaled by 2^-kReluShift (dev_params.h: the relu is the
  // VALU's [0, 1] clamp on a packed add): the input layer's weights and every bias row of
  // the tower carry `dn`, the output layer's weights `up`.  Exact: powers of two.
  const bool relu_clamp = dp.act == ddd::ACT_RELU && ddd::kReluShift != 0;
  const float dn = relu_clamp ? std::ldexp(1.0f, -ddd::kReluShift) : 1.0f;
  const float up = relu_clamp ? std::ldexp(1.0f, ddd::kReluShift) : 1.0f;
  if (m->big()) {
    // streamed layouts of rhs_mfma.h: input_layer_big / hidden_layer_stream
    const int in_steps = (tk + 2) / 2;
    {
      const float* w = weights + net.w_off[0];   // [tk][1][tc]
      const float* b = weights + net.b_off[0];
      std::vector<float> packed((size_t)tcb * in_steps * 64, 0.0f);
      for (int h = 0; h < tcb; ++h)
        for (int s = 0; s < in_steps; ++s)
          for (int lane = 0; lane < 64; ++lane) {
            const int k = 2 * s + (lane >> 5), ch = 32 * h + (lane & 31);
            packed[((size_t)h * in_steps + s) * 64 + lane] =
                dn * (k < tk ? w[k * tc + ch] : k == tk ? b[ch] : 0.0f);
          }
      int rc = upload(packed, &m->d_w_input);
      if (rc) return rc;
      m->dp.w_input = m->d_w_input;
    }
    if (hidden > 0) {
      const int groups = tk * tc / 8;   // Tower::kHidGroups
      const size_t layer_floats = (size_t)groups * tcb * 64 * 4 + (size_t)tcb * 64;
      std::vector<float> packed((size_t)hidden * layer_floats, 0.0f);
      for (int l = 0; l < hidden; ++l) {
        const float* w = weights + net.w_off[l + 1];   // [tk][tc][tc]
        const float* b = weights + net.b_off[l + 1];
        float* dst = packed.data() + (size_t)l * layer_floats;
        for (int g = 0; g < groups; ++g)
          for (int h = 0; h < tcb; ++h)
            for (int lane = 0; lane < 64; ++lane)
              for (int e = 0; e < 4; ++e) {
                const int s = 4 * g + e;                       // reduction step of this output block
                const int tap = s / (16 * tcb), cb = (s / 16) % tcb, jj = s % 16;
                const int cin = 32 * cb + 16 * (lane >> 5) + jj, cout = 32 * h + (lane & 31);
                dst[(((size_t)g * tcb + h) * 64 + lane) * 4 + e] =
                    w[((size_t)tap * tc + cin) * tc + cout];
              }
        float* bias = dst + (size_t)groups * tcb * 64 * 4;
        for (int h = 0; h < tcb; ++h)
          for (int lane = 0; lane < 32; ++lane) bias[h * 64 + lane] = dn * b[32 * h + lane];
      }
      int rc = upload(packed, &m->d_w_hidden);
      if (rc) return rc;
      m->dp.w_hidden = m->d_w_hidden;
    }
  } else {
  {
    // input layer 1 -> 32: k = 2 s + (lane >> 5) is the tap, k = 5 the bias
    const float* w = weights + net.w_off[0];   // [5][1][32]
    const float* b = weights + net.b_off[0];
    std::vector<float> packed((size_t)ddd::mfma::kInSteps * 64, 0.0f);
    for (int s = 0; s < ddd::mfma::kInSteps; ++s)
      for (int lane = 0; lane < 64; ++lane) {
        const int k = 2 * s + (lane >> 5), ch = lane & 31;
        packed[s * 64 + lane] = dn * (k < 5 ? w[k * 32 + ch] : b[ch]);
      }
    int rc = upload(quad_rows(packed.data(), ddd::mfma::kInSteps), &m->d_w_input);
    if (rc) return rc;
    m->dp.w_input = m->d_w_input;
  }
  if (hidden > 0) {
    std::vector<float> packed((size_t)hidden * ddd::mfma::kHidSteps * 64, 0.0f);
    for (int h = 0; h < hidden; ++h) {
      const float* w = weights + net.w_off[h + 1];   // [5][32][32]
      const float* b = weights + net.b_off[h + 1];
      float* dst = packed.data() + (size_t)h * ddd::mfma::kHidSteps * 64;
      for (int s = 0; s < 80; ++s) {
        const int tap = s / 16, jj = s % 16;
        for (int lane = 0; lane < 64; ++lane) {
          const int cin = 16 * (lane >> 5) + jj, cout = lane & 31;
          dst[s * 64 + lane] = w[(tap * 32 + cin) * 32 + cout];
        }
      }
      for (int lane = 0; lane < 64; ++lane)
        dst[80 * 64 + lane] = (lane >> 5) == 0 ? dn * b[lane & 31] : 0.0f;
    }
    std::vector<float> stored;   // every hidden layer padded on its own (load_hidden)
    for (int h = 0; h < hidden; ++h) {
      const std::vector<float> q =
          quad_rows(packed.data() + (size_t)h * ddd::mfma::kHidSteps * 64, ddd::mfma::kHidSteps);
      stored.insert(stored.end(), q.begin(), q.end());
    }
    int rc = upload(stored, &m->d_w_hidden);
    if (rc) return rc;
    m->dp.w_hidden = m->d_w_hidden;
  }
  }   // default tower
  {
    const int l = dp.L - 1;
    const int kc = tk * tc;                      // reduction length of the output layer
    const int fin_k = kc + 1;                    // ... + the bias row (Tower::kFinK)
    const auto fin_regs = [fin_k](int groups) { return (fin_k * groups + 15) / 16; };
    const float* w_nat = weights + net.w_off[l];   // [5][32][C_out]
    const float* b_nat = weights + net.b_off[l];
    // Fold coeff = bias + net[start:stop] @ nullspace into the output layer:
    // W'[tap][cin][G d + g] = sum_j W[tap][cin][start_d + j] * ns_d[j][g]
    // (accumulated in double, rounded once to float32), same for the bias.
    // The layer then emits the D x G coefficients directly and the epilogue's
    // projection disappears.  Deviation from the reference's operation order:
    // O(1 ulp) of the coefficient deltas, far inside the 1e-5 tolerance.
    // (wf / bf: 16 columns, channel G d + g; D <= 2, G <= 8.)
    std::vector<float> wf, bf;
    const bool projected = dp.target == ddd::TARGET_COEFFICIENTS && dp.pao > 0;
    const bool direct_coeffs = dp.target == ddd::TARGET_COEFFICIENTS && dp.pao <= 0;
    // the kernels' folded epilogue exists for two derivatives and 6..8 stencil points
    // (default flavour: channel G d + g of 16) and -- always -- for the wide flavour's
    // coefficient nets (channel wide_slot(G) d + g of 36, up to three derivatives: the wide
    // kernels have no projection code at all)
    const bool wide_fold = m->wide && dp.target == ddd::TARGET_COEFFICIENTS;
    const int slot = wide_fold ? ddd::mfma::wide_slot(dp.G) : dp.G;
    const int fold_cols = wide_fold ? ddd::mfma::flavour_net_channels(true) : 16;
    const bool fold_shape = wide_fold ? dp.D <= ddd::mfma::kWideDerivs
                                      : dp.D <= 2 && dp.G >= 6 && dp.G <= ddd::kGMax && !m->wide;
    bool can_fold = projected && fold_shape && (!g_debug.no_fold || wide_fold);
    if (direct_coeffs && fold_shape) {
      // the net emits the D x G coefficients themselves (model.py:460-475) in
      // exactly the folded layer's channel order: nothing to project
      can_fold = true;
      wf.assign((size_t)kc * fold_cols, 0.0f);
      bf.assign(fold_cols, 0.0f);
      for (int d = 0; d < dp.D; ++d)
        for (int g = 0; g < dp.G; ++g) {
          const int c = dp.G * d + g, oc = slot * d + g;
          for (int row = 0; row < kc; ++row)
            wf[(size_t)row * fold_cols + oc] = w_nat[(size_t)row * dp.C_out + c];
          bf[oc] = b_nat[c];
        }
    } else if (can_fold) {
      wf.assign((size_t)kc * fold_cols, 0.0f);
      bf.assign(fold_cols, 0.0f);
      for (int d = 0; d < dp.D; ++d)
        for (int g = 0; g < dp.G; ++g) {
          const int oc = slot * d + g;
          for (int row = 0; row < kc; ++row) {
            double acc = 0.0;
            for (int j = 0; j < dp.in_size[d]; ++j)
              acc += (double)w_nat[(size_t)row * dp.C_out + dp.in_start[d] + j] *
                     (double)dp.ns8[dp.in_start[d] + j][g];
            wf[(size_t)row * fold_cols + oc] = (float)acc;
          }
          // bias row of the folded layer: the accuracy layer's standard
          // coefficients + the projected conv bias, rounded once
          double acc = (double)dp.bias8[d][g];
          for (int j = 0; j < dp.in_size[d]; ++j)
            acc += (double)b_nat[dp.in_start[d] + j] * (double)dp.ns8[dp.in_start[d] + j][g];
          bf[oc] = (float)acc;
        }
    }
    // Packing for the 4x4x1 broadcast MFMA (rhs_mfma.h: final_layer4): channels
    // grouped by four; instruction q = k * groups + grp reads lanes
    // 4 (q % 16) .. + 3 of weight register q / 16, lane 4 abid + r carrying
    // channel 4 grp + r; k = (tap, cin) in natural order, k = 160: bias.
    // `folded`: source = the folded layer (16 columns, channel G d + g), else the
    // natural one; `renumber`: only the live channels.
    auto pack4 = [&](int groups, bool folded, bool renumber) {
      const float* w = folded ? wf.data() : w_nat;
      const float* b = folded ? bf.data() : b_nat;
      const int cout_n = folded ? fold_cols : dp.C_out;
      const int n_ch = folded ? dp.D * dp.G : dp.C_out;
      std::vector<float> packed4((size_t)ddd::mfma::fin4_regs(4) * 64, 0.0f);
      for (int k = 0; k < ddd::mfma::kFin4K; ++k)
        for (int grp = 0; grp < groups; ++grp) {
          const int q = k * groups + grp;
          for (int r = 0; r < 4; ++r) {
            const int ch = 4 * grp + r;
            int src = ch;
            if (renumber && ch >= n_ch) continue;   // folded columns are contiguous already
            if (src >= cout_n) continue;
            packed4[(size_t)(q / 16) * 64 + 4 * (q % 16) + r] =
                k < 160 ? up * w[(size_t)k * cout_n + src] : b[src];
          }
        }
      return packed4;
    };
    // Run-time-parameterised kernels: only the live channel groups are issued,
    // as interleaved accumulator chains (rhs_mfma.h: two or three chains run at
    // 8.1 cycles per MFMA, a lone group at 13.2).  Folding the projection trades the
    // epilogue's ~C_out x G FMAs (~2.5 units of 161 MFMA slots) for D x G
    // instead of C_out channels: fold only where the matrix work does not grow
    // by more than that (the same outcome as rhs_mfma.h: spec_folded for the six
    // default models, so the two kernel families stay bit-identical).
    // polynomial_accuracy_order = 0 has nothing to project.
    const auto issue_cost = [](int groups) { return groups == 1 ? 13.2 : 8.1 * groups; };
    const int groups_rt_plain = (dp.C_out + 3) / 4;
    const int groups_rt_folded = ((dp.D - 1) * slot + dp.G + 3) / 4;
    const bool fold_rt = can_fold && (direct_coeffs || wide_fold ||
                                      issue_cost(groups_rt_folded) <=
                                          issue_cost(groups_rt_plain) + 2.5);
    if (wide_fold && !fold_rt) return DDD_ERR_UNSUPPORTED;   // (decide_mfma admits only what folds)
    m->dp.folded = fold_rt ? 1 : 0;
    m->dp.rt_groups = fold_rt ? groups_rt_folded : groups_rt_plain;
    {
      const float* w = fold_rt ? wf.data() : w_nat;
      const float* b = fold_rt ? bf.data() : b_nat;
      const int cout_n = fold_rt ? fold_cols : dp.C_out;
      // layout: the head chunk (rt_head_groups: 0, 1 or 3 groups, fin4_regs(head)
      // rows), then the pairs (fin4_regs(2) rows each); + slack so that the
      // kernels' fixed-size first fetch (fin4_regs(3) rows) stays inside
      const int groups = m->dp.rt_groups, head = ddd::mfma::rt_head_groups(groups);
      const int pair_rows = fin_regs(2), head_rows = fin_regs(head);
      const int total_rows = head_rows + (groups - head) / 2 * pair_rows + fin_regs(3);
      std::vector<float> packed((size_t)total_rows * 64, 0.0f);
      const auto pack_chunk = [&](int row0, int first_group, int ng) {
        for (int k = 0; k < fin_k; ++k)
          for (int gi = 0; gi < ng; ++gi) {
            const int q = k * ng + gi;
            for (int r = 0; r < 4; ++r) {
              const int ch = 4 * (first_group + gi) + r;
              if (ch >= cout_n) continue;
              packed[((size_t)row0 + q / 16) * 64 + 4 * (q % 16) + r] =
                  k < kc ? up * w[(size_t)k * cout_n + ch] : b[ch];
            }
          }
      };
      if (head > 0) pack_chunk(0, 0, head);
      for (int g0 = head; g0 < groups; g0 += 2)
        pack_chunk(head_rows + (g0 - head) / 2 * pair_rows, g0, 2);
      int rc2 = upload(packed, &m->d_w_final4_rt);
      if (rc2) return rc2;
      m->dp.w_final4_rt = m->d_w_final4_rt;
    }
    int rc = DDD_OK;
    // specialised kernels: live channels only; folded only where that does not
    // cost a channel group (same rule as rhs_mfma.h: spec_folded)
    const int groups_folded = (dp.D * dp.G + 3) / 4, groups_plain = (dp.C_out + 3) / 4;
    m->spec_folded = can_fold && groups_folded <= groups_plain;
    m->dp.fin4_groups = m->spec_folded ? groups_folded : groups_plain;
    if (!m->wide && !m->big()) {   // (the specialised kernels: default tower, not wide)
      rc = upload(quad_rows(pack4(m->dp.fin4_groups, m->spec_folded, true).data(),
                            ddd::mfma::fin4_regs(4)),
                  &m->d_w_final4);
      if (rc) return rc;
      m->dp.w_final4 = m->d_w_final4;
      // the same layer for the split integrators (rhs_mfma.h kSplit): two chunks of
      // channel groups, each packed on its own and quad-stored in 24 rows
      {
        const float* w = m->spec_folded ? wf.data() : w_nat;
        const float* b = m->spec_folded ? bf.data() : b_nat;
        const int cout_n = m->spec_folded ? fold_cols : dp.C_out;
        const int n_ch = m->spec_folded ? dp.D * dp.G : dp.C_out;
        const int groups = m->dp.fin4_groups, na = (groups + 1) / 2;
        const int chunk_rows = ddd::mfma::fin4_regs(2);
        std::vector<float> both;
        for (int c = 0; c < 2; ++c) {
          const int g0 = c == 0 ? 0 : na, ng = c == 0 ? na : groups - na;
          std::vector<float> chunk((size_t)chunk_rows * 64, 0.0f);
          for (int k = 0; k < ddd::mfma::kFin4K && ng > 0 && ng <= 2; ++k)
            for (int gi = 0; gi < ng; ++gi) {
              const int q = k * ng + gi;
              for (int r = 0; r < 4; ++r) {
                const int ch = 4 * (g0 + gi) + r;
                if (ch >= n_ch || ch >= cout_n) continue;
                chunk[(size_t)(q / 16) * 64 + 4 * (q % 16) + r] =
                    k < 160 ? up * w[(size_t)k * cout_n + ch] : b[ch];
              }
            }
          const std::vector<float> q4 = quad_rows(chunk.data(), chunk_rows);
          both.insert(both.end(), q4.begin(), q4.end());
        }
        if (groups <= 4) {
          rc = upload(both, &m->d_w_final4_split);
          if (rc) return rc;
          m->dp.w_final4_split = m->d_w_final4_split;
        }
      }
      // ... and the whole net for the integrators on FOUR 16-row wavefronts (rhs_mfma.h
      // kQuad), every layer as v_mfma_f32_16x16x4_f32 A operands: lane l supplies
      // W[out = l & 15][reduction slot sg = l >> 4] of a step.  Three-layer nets only
      // (the per-equation kernels'), <= 16 output channels in the renumbering of w_final4.
      if (dp.L == 3 && m->dp.fin4_groups <= 4) {
        using namespace ddd::mfma;
        const float* w = m->spec_folded ? wf.data() : w_nat;
        const float* b = m->spec_folded ? bf.data() : b_nat;
        const int cout_n = m->spec_folded ? fold_cols : dp.C_out;
        const int n_ch = m->spec_folded ? dp.D * dp.G : dp.C_out;
        std::vector<float> quad((size_t)kQuadRows * 64, 0.0f);
        const float* w0 = weights + net.w_off[0];   // [5][1][32]
        const float* b0 = weights + net.b_off[0];
        const float* w1 = weights + net.w_off[1];   // [5][32][32]
        const float* b1 = weights + net.b_off[1];
        for (int chh = 0; chh < 2; ++chh)
          for (int lane = 0; lane < 64; ++lane) {
            const int sg = lane >> 4, cout = 16 * chh + (lane & 15);
            // input layer: step 0 = taps 0..3, step 1 = tap 4, bias, 0, 0 (input_layer's k order)
            quad[(size_t)(chh * kQuadInSteps + 0) * 64 + lane] = dn * w0[sg * 32 + cout];
            quad[(size_t)(chh * kQuadInSteps + 1) * 64 + lane] =
                sg == 0 ? dn * w0[4 * 32 + cout] : sg == 1 ? dn * b0[cout] : 0.0f;
            // hidden layer: step 8 tap + i, slot sg -> cin = (sg >> 1) + 16 (sg & 1) + 2 i:
            // per tap c = 0, 16, 1, 17, ... -- hidden_layer's order (s = 16 tap + jj, half = l >> 5)
            float* hid = quad.data() + (size_t)(2 * kQuadInSteps + chh * kQuadHidSteps) * 64;
            for (int s2 = 0; s2 < 40; ++s2) {
              const int tap = s2 / 8, i = s2 % 8;
              const int cin = (sg >> 1) + 16 * (sg & 1) + 2 * i;
              hid[(size_t)s2 * 64 + lane] = w1[(tap * 32 + cin) * 32 + cout];
            }
            hid[(size_t)40 * 64 + lane] = sg == 0 ? dn * b1[cout] : 0.0f;
          }
        // output layer: step s2, slot sg -> k = 4 s2 + sg in natural order (final_layer4's), k = 160: bias
        float* fin = quad.data() + (size_t)(2 * kQuadInSteps + 2 * kQuadHidSteps) * 64;
        for (int s2 = 0; s2 < kQuadFinSteps; ++s2)
          for (int lane = 0; lane < 64; ++lane) {
            const int k = 4 * s2 + (lane >> 4), ch = lane & 15;
            if (ch >= n_ch || ch >= cout_n || k > 160) continue;
            fin[(size_t)s2 * 64 + lane] = k < 160 ? up * w[(size_t)k * cout_n + ch] : b[ch];
          }
        rc = upload(quad, &m->d_w_quad);
        if (rc) return rc;
        m->dp.w_quad = m->d_w_quad;
      }
      // ... and, for nets of up to 16 filters (embedded here in 32), the BLOCK-DIAGONAL packing
      // of rhs_mfma.h HalfTower: the hidden layer's panel with output rows 16..31 = the same 16
      // channels fed by reduction half 1, and the output layer over 5 x 16 + 1 reduction steps.
      // Source: the embedded net (channels >= 16 are zero there).
      if (dp.L == 3 && dp.cout[0] <= 16 && m->dp.fin4_groups <= 4) {
        using namespace ddd::mfma;
        const float* w1 = weights + net.w_off[1];   // [5][32][32]
        const float* b1 = weights + net.b_off[1];
        std::vector<float> hid((size_t)kHidSteps * 64, 0.0f);
        for (int s2 = 0; s2 < 80; ++s2) {
          const int tap = s2 / 16, jj = s2 % 16;
          for (int lane = 0; lane < 64; ++lane) {
            const int mrow = lane & 31, half = lane >> 5;
            if ((mrow < 16) == (half == 0))
              hid[(size_t)s2 * 64 + lane] = w1[(tap * 32 + jj) * 32 + (mrow & 15)];
          }
        }
        for (int lane = 0; lane < 32; ++lane) hid[(size_t)80 * 64 + lane] = dn * b1[lane & 15];
        rc = upload(quad_rows(hid.data(), kHidSteps), &m->d_w_hidden_half);
        if (rc) return rc;
        const float* w = m->spec_folded ? wf.data() : w_nat;
        const float* b = m->spec_folded ? bf.data() : b_nat;
        const int cout_n = m->spec_folded ? fold_cols : dp.C_out;
        const int n_ch = m->spec_folded ? dp.D * dp.G : dp.C_out;
        const int groups = m->dp.fin4_groups;
        std::vector<float> fin((size_t)fin4_regs(4) * 64, 0.0f);
        for (int k = 0; k <= 80; ++k)   // k = 16 tap + cin, 80: bias
          for (int grp = 0; grp < groups; ++grp) {
            const int q = k * groups + grp;
            for (int r = 0; r < 4; ++r) {
              const int ch = 4 * grp + r;
              if (ch >= n_ch || ch >= cout_n) continue;
              fin[(size_t)(q / 16) * 64 + 4 * (q % 16) + r] =
                  k < 80 ? up * w[(size_t)((k / 16) * 32 + (k % 16)) * cout_n + ch] : b[ch];
            }
          }
        rc = upload(quad_rows(fin.data(), fin4_regs(4)), &m->d_w_final4_half);
        if (rc) return rc;
        // ... and as A operands of the 16x16x4 layers (Tile16Tower): lane l = W[out = l & 15][slot l >> 4];
        // input layer: step 0 = taps 0..3, step 1 = tap 4, bias, 0, 0; hidden layer: step 4 tap + e,
        // slot sg -> input channel 4 e + sg; step 20: the bias in slot 0
        const float* w0 = weights + net.w_off[0];   // [5][1][32]
        const float* b0 = weights + net.b_off[0];
        std::vector<float> t16((size_t)(kT16InSteps + kT16HidSteps) * 64, 0.0f);
        for (int lane = 0; lane < 64; ++lane) {
          const int sg = lane >> 4, cout = lane & 15;
          t16[(size_t)0 * 64 + lane] = dn * w0[sg * 32 + cout];
          t16[(size_t)1 * 64 + lane] = sg == 0 ? dn * w0[4 * 32 + cout] : sg == 1 ? dn * b0[cout] : 0.0f;
          for (int s2 = 0; s2 < 20; ++s2) {
            const int tap = s2 / 4, e = s2 % 4;
            t16[(size_t)(kT16InSteps + s2) * 64 + lane] = w1[(tap * 32 + 4 * e + sg) * 32 + cout];
          }
          t16[(size_t)(kT16InSteps + 20) * 64 + lane] = sg == 0 ? dn * b1[cout] : 0.0f;
        }
        rc = upload(t16, &m->d_w_t16);
        if (rc) return rc;
      }
    }
  }
  return DDD_OK;
}

// Nets between two towers ride the next tower up EXACTLY (rhs_mfma.h: Tower; the
// default one has 5 taps x 32 channels), embedded with
// zero weights: a K-tap kernel (K < 5) is the 5-tap kernel whose outer taps are
// zero (tap k of K sits at offset k - ceil((K-1)/2), the alignment of
// layers.pad_periodic(center=True), layers.py:76-79), F < 32 filters are 32
// filters whose extra rows / columns / biases are zero.  fma(0, x, acc) == acc
// for every finite x, and a padded channel is multiplied by zero weights in the
// next layer whatever the activation makes of its 0, so the finite results are
// bit-identical to the unpadded evaluation order-for-order; the matrix work
// grows by 5/K and (32/F)^2, still an order of magnitude ahead of the generic
// kernel.  (Algorithmic FLOPs -- ddd_fma_per_point -- keep counting the true net.)
void embed_tower(const ddd::DevParams& dp, const std::vector<float>& wv, int tower_k,
                 int tower_c, std::vector<float>* padded, NetLayout* net) {
  const int k5 = tower_k, f32 = tower_c;       // (the tower's taps and filters)
  const int shift = (k5 - 1) / 2 - dp.K / 2;   // ceil((k5-1)/2) - ceil((K-1)/2), k5 odd
  padded->clear();
  for (int l = 0; l < dp.L; ++l) {
    const int cin = l == 0 ? 1 : f32;
    const int cout = l == dp.L - 1 ? dp.C_out : f32;
    net->w_off[l] = (int)padded->size();
    padded->resize(padded->size() + (size_t)k5 * cin * cout, 0.0f);
    net->b_off[l] = (int)padded->size();
    padded->resize(padded->size() + (size_t)cout, 0.0f);
    const float* w = wv.data() + dp.w_off[l];
    const float* b = wv.data() + dp.b_off[l];
    for (int k = 0; k < dp.K; ++k)
      for (int ci = 0; ci < dp.cin[l]; ++ci)
        for (int co = 0; co < dp.cout[l]; ++co)
          (*padded)[(size_t)net->w_off[l] + ((size_t)(k + shift) * cin + ci) * cout + co] =
              w[((size_t)k * dp.cin[l] + ci) * dp.cout[l] + co];
    for (int co = 0; co < dp.cout[l]; ++co) (*padded)[(size_t)net->b_off[l] + co] = b[co];
  }
  net->weights = padded->data();
}

// One-layer nets the MFMA-path kernels carry on their VALU route (DevParams::linear_taps).
bool linear_eligible(const ddd::DevParams& dp) {
  return !dp.fixed && dp.L == 1 && dp.target == ddd::TARGET_COEFFICIENTS &&
         !(dp.pao <= 0 && dp.unbiased) && dp.D <= 3 && dp.G <= ddd::kGMax && dp.K <= 7 &&
         dp.K * dp.D <= ddd::kChMax;
}

void decide_mfma(ddd_model* m) {
  const ddd::DevParams& dp = m->dp;
  char why[256] = "";
  bool ok = true;
  auto no = [&](const char* msg) { if (ok) snprintf(why, sizeof(why), "%s", msg); ok = false; };
  if (dp.N < 8 || dp.N > 256) no("num_points outside [8, 256]");
  if (dp.G > ddd::kGWide) no("stencil wider than 12");
  if (dp.G > dp.N) no("stencil wider than the grid");
  if (dp.fixed && dp.weno) no("WENO reconstruction");
  // The wide flavour of the run-time-parameterised kernels (stencils up to 12
  // points, up to 24 net output channels, projection folded into the output layer) carries what
  // the default flavour (8 points, 16 channels) cannot: coefficient_grid_min_size
  // = 9 and polynomial_accuracy_order = 0 with three derivatives
  // (training_test.py:56-57).
  bool wide = dp.G > ddd::kGMax;
  const bool linear = linear_eligible(dp);
  if (!dp.fixed && !linear) {
    // direct heads (space_derivatives / time_derivative / flux: D or 1 output
    // channels) and polynomial_accuracy_order = 0 (D x G coefficient channels,
    // no projection) run on the run-time-parameterised MFMA kernels
    if (dp.target == ddd::TARGET_COEFFICIENTS && dp.pao <= 0 && dp.D > 2) wide = true;
    if (dp.C_out > ddd::kChMax) wide = true;
    // the smallest tower built that holds the net (launch.h: DDD_FOR_EACH_BIG_TOWER + the
    // default 5 x 32); nets in between are packed zero-padded (embed_tower)
    m->tower_k = dp.K <= 3 ? 3 : dp.K <= 5 ? 5 : 7;
    m->tower_cb = dp.F <= 32 ? 1 : 2;
    if (m->tower_cb == 2 && m->tower_k == 3) m->tower_k = 5;   // (64-filter towers: 5 and 7 taps)
    // up to 16 filters of up to 3 taps in the architecture of the per-equation kernels: the
    // 5 x 32 tower, whose persistent integrators carry such nets block-diagonally (rhs_mfma.h
    // HalfTower: half the hidden-layer MFMAs of the embedding) -- ahead of the 3-tap tower,
    // where they would occupy a quarter of every MFMA
    if (m->tower_k == 3 && dp.F <= 16 && dp.L == 3 && dp.act == ddd::ACT_RELU && !wide &&
        dp.target == ddd::TARGET_COEFFICIENTS && dp.pao > 0 &&
        dp.equation >= ddd::EQ_BURGERS && dp.equation <= ddd::EQ_KS_CONS &&
        dp.G == ddd::mfma::spec_stencil(dp.equation))
      m->tower_k = 5;
    if (dp.F > 64) no("filter_size > 64");
    if (dp.K > 7) no("kernel_size > 7");
    if (dp.L < 2) no("fewer than 2 conv layers");
    // the wide flavour folds every coefficient net into its output layer (pack_mfma_weights):
    // three derivatives' worth of register slots; the projection tables hold 24 net channels
    if (wide && dp.target == ddd::TARGET_COEFFICIENTS && dp.D > ddd::mfma::kWideDerivs)
      no("wide stencils / > 16 output channels with four derivatives");
    if (dp.C_out > ddd::kChWide && !(wide && dp.target == ddd::TARGET_COEFFICIENTS && dp.pao <= 0))
      no("more than 24 output channels");
    // stencils > 8 points / > 16 channels exist on the 5 x 32 tower only (wide flavour)
    if (wide && m->tower_k == 3) m->tower_k = 5;
    if (wide && m->big()) no("wide stencils / > 16 output channels with a tower other than 5 x 32");
  }
  if (!ok || dp.fixed || linear) { m->tower_k = 5; m->tower_cb = 1; }
  m->wide = ok && wide;
  m->mfma_ok = ok;
  m->mfma_reason = why;
  m->kernel = ok ? DDD_KERNEL_MFMA : DDD_KERNEL_GENERIC;
}

int check_generic_lds(const ddd_model* m, int state_bytes) {
  const size_t need = ddd::generic::lds_bytes(m->dp, state_bytes);
  if (need > 160 * 1024)
    return fail(DDD_ERR_UNSUPPORTED,
                "generic kernel needs %zu B of LDS per sample (limit 163840): "
                "num_points x channels too large", need);
  return DDD_OK;
}

int ensure_scratch(ddd_model* m, size_t floats) {
  if (m->scratch_floats >= floats) return DDD_OK;
  free_dev(m->d_scratch);
  m->d_scratch = nullptr;
  m->scratch_floats = 0;
  DDD_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_scratch), floats * sizeof(float)));
  m->scratch_floats = floats;
  return DDD_OK;
}

int check_batch(const ddd_model* m, int batch, bool want_spectral = false) {
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  if (m->spectral != want_spectral)
    return fail(DDD_ERR_UNSUPPORTED,
                m->spectral ? "spectral (float64) models only accept the *_f64 entry points"
                            : "this entry point needs a spectral model (ddd_spectral_create)");
  if (batch < 0) return fail(DDD_ERR_INVALID_ARGUMENT, "negative batch");
  if (m->dp.forced && batch > m->dp.forcing_batch)
    return fail(DDD_ERR_INVALID_ARGUMENT,
                "batch %d exceeds the %d samples forcing was set for", batch,
                m->dp.forcing_batch);
  return DDD_OK;
}

// Geometry of the MFMA path.  rows: grid points per workgroup; wave_rows: per
// wavefront.  One free-running 64-row wavefront per workgroup when whole
// samples fit 64 rows (optionally split over two 32-row wavefronts), else 256
// rows on four wavefronts with block barriers.
struct MfmaGeometry { int rows, wave_rows; };

// One-time hardware check of the DPP wavefront rotate the one-wave kernel uses
// for the flux exchange when N = 64 (falls back to ds_bpermute otherwise).
int dpp_wave_rol_ok() {
  static int ok = -1;
  if (ok >= 0) return ok;
  ok = 0;
  float* d = nullptr;
  if (hipMalloc(reinterpret_cast<void**>(&d), 64 * sizeof(float)) != hipSuccess) return ok;
  hipLaunchKernelGGL(ddd::ops::dpp_rotate_probe_kernel, dim3(1), dim3(64), 0, nullptr, d);
  float h[64];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
    ok = 1;
    for (int l = 0; l < 64; ++l)
      if (h[l] != (float)(((l + 1) % 64) * 3 + 1)) ok = 0;
  }
  (void)hipFree(d);
  return ok;
}

int device_simds() {
  static int simds = 0;
  if (simds == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      simds = prop.multiProcessorCount * 4;
    if (simds <= 0) simds = 1024;
  }
  return simds;
}

MfmaGeometry mfma_geometry(const ddd_model* m, int batch) {
  const bool fits64 = m->dp.N <= 64 && 64 % m->dp.N == 0;
  if (m->force_rows == 256 || !fits64) return {256, 64};
  if (m->force_rows == 64) return {64, 64};
  if (m->force_rows == 32 && !m->wide && !m->big()) return {64, 32};   // (no wide / other-tower split)
  if (m->force_rows == 16 && !m->wide && !m->big()) return {64, 16};   // (launch_integrate alone honours it)
  // Two 32-row wavefronts per sample are never the geometry of the fused substep or
  // the adaptive kernels; launch_integrate alone switches small float32 ensembles to
  // the split integrators (rhs_mfma.h kSplit), where the measurement says it pays.
  (void)batch;
  return {64, 64};
}

// Equation id of the compile-time specialised integrator this model can use,
// or -1: the default architecture (three relu conv layers, default stencil
// width, projection folded when D <= 2) on a non-Godunov equation.
int spec_equation(const ddd_model* m, int rows) {
  const ddd::DevParams& dp = m->dp;
  if (g_debug.no_spec || m->wide || m->big()) return -1;
  if (dp.forced) {
    // the specialised kernels only carry the harmonic-sum forcing (rhs_mfma.h:
    // launch_setup `fast`); exotic tables go to the run-time kernels
    const int spg = rows / dp.N;
    const bool fast = spg * dp.P <= rows && dp.n_k <= 6 &&
                      spg * ddd::mfma::kTrigMax <= 3 * rows / 4 && dp.P < 256;
    if (!fast) return -1;
  }
  if (dp.fixed || dp.L != 3 || dp.act != ddd::ACT_RELU) return -1;
  if (dp.target != ddd::TARGET_COEFFICIENTS || dp.pao <= 0) return -1;
  if (dp.equation < ddd::EQ_BURGERS || dp.equation > ddd::EQ_KS_CONS) return -1;
  if (dp.D != ddd::mfma::spec_derivs(dp.equation)) return -1;
  if (dp.G != ddd::mfma::spec_stencil(dp.equation)) return -1;
  if ((dp.conservative != 0) != ddd::mfma::spec_flux_form(dp.equation)) return -1;
  // the kernels hard-wire the null-space split (rhs_mfma.h: spec_in_size) and
  // whether their packed output layer is folded
  int start = 0;
  for (int d = 0; d < dp.D; ++d) {
    if (dp.in_size[d] != ddd::mfma::spec_in_size(dp.equation, d) || dp.in_start[d] != start)
      return -1;
    start += dp.in_size[d];
  }
  if (m->spec_folded != ddd::mfma::spec_folded(dp.equation)) return -1;
  return dp.equation;
}

bool aligned16(const void* ptr) {
  return (reinterpret_cast<uintptr_t>(ptr) & 15u) == 0;
}

// Fixed-stencil models without forcing: the streaming kernel (rhs_stream.h).
bool use_stream_kernel(const ddd_model* m, const ddd::SubstepArgs& a) {
  if (!ddd::stream::supports(m->dp) || m->explicit_kernel) return false;
  if (a.derivs_out != nullptr || a.coeffs_out != nullptr) return false;
  if (g_debug.no_stream) return false;
  return aligned16(a.y_in) && aligned16(a.y_base) && aligned16(a.y_out) &&
         aligned16(a.acc_in) && aligned16(a.acc_out);
}

// WENO5 + Godunov-flux models (integrate.WENODifferentiator, the exact Burgers solver) on
// the one-wavefront-per-sample kernels of rhs_weno.h; an explicit ddd_set_kernel(GENERIC)
// keeps the generic kernel (A/B, and the grids rhs_weno.h does not carry).
bool use_weno_kernel(const ddd_model* m) {
  return m->kernel == DDD_KERNEL_GENERIC && !m->explicit_kernel && !g_debug.no_weno &&
         ddd::weno::supports(m->dp);
}

// sample0: index of a.y_in's first sample in the model's per-sample tables (a
// half-ensemble launch); grid_share: this launch may occupy 1 / grid_share of
// the machine-sized grid (it runs next to grid_share - 1 others).
int launch_substep(ddd_model* m, const ddd::SubstepArgs& a, hipStream_t stream, int sample0 = 0,
                   int grid_share = 1) {
  if (a.batch == 0) return DDD_OK;
  m->last_batch = a.batch;
  if (use_stream_kernel(m, a)) {
    const long pts = (long)ddd::stream::samples_per_block(m->dp.N) * m->dp.N;
    const long total = (long)a.batch * m->dp.N;
    const unsigned blocks = (unsigned)((total + pts - 1) / pts);
    if (ddd::stream::quads_for(m->dp.N) == 2)
      hipLaunchKernelGGL(ddd::stream::fixed_substep_kernel<2>, dim3(blocks),
                         dim3(ddd::stream::kThreads), 0, stream, m->dp, a);
    else
      hipLaunchKernelGGL(ddd::stream::fixed_substep_kernel<1>, dim3(blocks),
                         dim3(ddd::stream::kThreads), 0, stream, m->dp, a);
    m->last_launch_streamed = true; m->last_launch_lean = false;
    DDD_HIP(hipGetLastError());
    return DDD_OK;
  }
  m->last_launch_streamed = false; m->last_launch_lean = false;
  m->last_launch_split = false; m->last_launch_quad = false; m->last_launch_half = false;
  if (m->kernel == DDD_KERNEL_MFMA) {
    m->dp.dpp_rol = dpp_wave_rol_ok();
    ddd::DevParams dp = m->dp;
    if (sample0 != 0 && dp.forced) {   // per-sample forcing rows of this slab
      dp.frc += (size_t)sample0 * dp.P;
      dp.runs += (size_t)sample0 * 8;
    }
    MfmaGeometry geo = mfma_geometry(m, a.batch);
    if (geo.wave_rows == 16) geo = {64, 64};   // (four-wavefront groups: persistent integrators only)
    const int spg = geo.rows / m->dp.N;
    const int blocks = (a.batch + spg - 1) / spg;
    // per-equation instantiations for the plain substep (no derivative views)
    const int eq = (a.derivs_out == nullptr && a.coeffs_out == nullptr && geo.wave_rows == 64)
                       ? spec_equation(m, geo.rows) : -1;
    // specialised models: machine-sized grid, weights resident per wavefront,
    // each group walks over several row groups (substep_multi_kernel)
    const int grid = std::min(blocks, (geo.rows == 64 ? 2 * device_simds() : device_simds() / 2) /
                                          std::max(grid_share, 1));
    // nets of up to 16 filters: 16-channel tiles (one-wave groups)
    const bool half = eq >= 0 && geo.rows == 64 && m->d_w_hidden_half != nullptr &&
                      m->d_w_final4_half != nullptr && m->d_w_t16 != nullptr && !g_debug.no_half;
    m->last_launch_half = half;
    if (half) { dp.w_hidden = m->d_w_hidden_half; dp.w_final4 = m->d_w_final4_half; dp.w_quad = m->d_w_t16; }
#define DDD_SUBSTEP_CASE(EQ) \
    case EQ:                                                                           \
      if (half) ddd::launch::substep_half_spec<EQ>(dp, a, blocks, grid, stream);       \
      else ddd::launch::substep_spec<EQ>(geo.rows, dp, a, blocks, grid, stream);       \
      break;
    switch (eq) {
      DDD_SUBSTEP_CASE(ddd::EQ_BURGERS)
      DDD_SUBSTEP_CASE(ddd::EQ_BURGERS_CONS)
      DDD_SUBSTEP_CASE(ddd::EQ_KDV)
      DDD_SUBSTEP_CASE(ddd::EQ_KDV_CONS)
      DDD_SUBSTEP_CASE(ddd::EQ_KS)
      DDD_SUBSTEP_CASE(ddd::EQ_KS_CONS)
      default:
        if (m->big()) {
#define DDD_BIG_SUBSTEP(K, CB)                                                          \
          if (m->tower_k == K && m->tower_cb == CB) {                                   \
            if (geo.rows == 64) ddd::launch::substep_big_unit<K, CB, 64>(dp, a, blocks, stream); \
            else ddd::launch::substep_big_unit<K, CB, 256>(dp, a, blocks, stream);      \
          }
          DDD_FOR_EACH_BIG_TOWER(DDD_BIG_SUBSTEP)
#undef DDD_BIG_SUBSTEP
        } else if (m->wide) {
          if (geo.rows == 64) ddd::launch::substep_wide_unit<64>(dp, a, blocks, stream);
          else ddd::launch::substep_wide_unit<256>(dp, a, blocks, stream);
        } else {
          ddd::launch::substep_runtime(geo.rows, geo.wave_rows, dp, a, blocks, stream);
        }
    }
#undef DDD_SUBSTEP_CASE
  } else if (use_weno_kernel(m) && a.coeffs_out == nullptr) {
    ddd::DevParams dp = m->dp;
    if (sample0 != 0 && dp.forced) {   // per-sample forcing rows of this slab
      dp.frc += (size_t)sample0 * dp.P;
      dp.runs += (size_t)sample0 * 8;
    }
    ddd::launch::weno_substep(dp, a, stream);
  } else {
    int rc = check_generic_lds(m, 0);
    if (rc) return rc;
    const size_t lds = ddd::generic::lds_bytes(m->dp, 0);
    DDD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ddd::generic::substep_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ddd::generic::substep_kernel, dim3(a.batch),
                       dim3(ddd::generic::kThreads), lds, stream, m->dp, a);
  }
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

template <int kRows, int kWR, typename ST>
void launch_mfma_integrate(ddd_model* m, const ddd::IntegrateArgs& a, hipStream_t stream) {
  m->dp.dpp_rol = dpp_wave_rol_ok();
  const int spg = kRows / m->dp.N;
  const int blocks = (a.batch + spg - 1) / spg;
  const bool hoist = !m->dp.fixed && m->dp.L == 3;
  constexpr bool f64 = std::is_same<ST, double>::value;
  // per-equation instantiations exist for 64-row wavefronts: float32 state in
  // both geometries, float64 state -- the SciPy-driven reference semantics,
  // integrate.py:154 -- in the one-wave geometry (launch.h)
  int eq = kWR == 64 ? spec_equation(m, kRows) : -1;   // (float64 state: both geometries since round 6)
  if (kWR == 32 && !f64 && m->dp.w_final4_split != nullptr) eq = spec_equation(m, kRows);
  if (kWR == 16) eq = (!f64 && m->dp.w_quad != nullptr) ? spec_equation(m, kRows) : -1;
  bool traced = false;
#ifdef DDD_PROBES
  if (a.trace != nullptr) {
    // phase tracing: the dedicated traced instantiation (headline config) or
    // the run-time-parameterised kernel
    traced = eq == ddd::EQ_BURGERS_CONS && kRows == 64 && !f64;
    if (!traced) eq = -1;
  }
#endif
  // nets of up to 16 filters: the block-diagonal tower (float32 state, one-wave groups)
  const bool half = kRows == 64 && kWR == 64 && !traced && eq >= 0 &&
                    m->d_w_hidden_half != nullptr && m->d_w_final4_half != nullptr && !g_debug.no_half;
  m->last_launch_half = half;
  ddd::DevParams dp_half = m->dp;
  if (half) { dp_half.w_hidden = m->d_w_hidden_half; dp_half.w_final4 = m->d_w_final4_half; dp_half.w_quad = m->d_w_t16; }
#define DDD_SPEC_CASE(EQ)                                                              \
  case EQ:                                                                             \
    if (half && f64) ddd::launch::integrate_half_f64_spec<EQ>(dp_half, a, blocks, stream); \
    else if (half) ddd::launch::integrate_half_spec<EQ>(dp_half, a, blocks, stream);   \
    else if (kWR == 16) ddd::launch::integrate_quad_spec<EQ>(m->dp, a, blocks, stream);     \
    else if (kWR == 32) ddd::launch::integrate_split_spec<EQ>(m->dp, a, blocks, stream); \
    else ddd::launch::integrate_spec<EQ>(kRows, f64, traced, m->dp, a, blocks, stream); \
    return;
  switch (eq) {
    DDD_SPEC_CASE(ddd::EQ_BURGERS)
    DDD_SPEC_CASE(ddd::EQ_BURGERS_CONS)
    DDD_SPEC_CASE(ddd::EQ_KDV)
    DDD_SPEC_CASE(ddd::EQ_KDV_CONS)
    DDD_SPEC_CASE(ddd::EQ_KS)
    DDD_SPEC_CASE(ddd::EQ_KS_CONS)
    default: break;
  }
#undef DDD_SPEC_CASE
  if (m->big()) {
    if constexpr (kWR == 64) {
#define DDD_BIG_INTEGRATE(K, CB)                                                         \
      if (m->tower_k == K && m->tower_cb == CB)                                          \
        ddd::launch::integrate_big_unit<K, CB, kRows, f64>(m->dp, a, blocks, stream);
      DDD_FOR_EACH_BIG_TOWER(DDD_BIG_INTEGRATE)
#undef DDD_BIG_INTEGRATE
    }
    return;
  }
  if (m->wide) {
    if (kWR == 64) ddd::launch::integrate_wide_unit<kRows, f64 ? 1 : 0>(hoist, m->dp, a, blocks,
                                                                        stream);
    return;
  }
  ddd::launch::integrate_runtime(kRows, kWR, f64, hoist, m->dp, a, blocks, stream);
}

// The persistent launch of fixed-stencil models and one-layer nets (rhs_lean.h): one
// instantiation per (taps, derivatives, stencil-column pairs).
template <int kK, int kD>
void launch_lean_kd(int pairs, const ddd::DevParams& dp, const ddd::IntegrateArgs& a, int blocks,
                    hipStream_t stream) {
  if (pairs == 3)
    hipLaunchKernelGGL((ddd::lean::integrate_kernel<kK, kD, 3>), dim3(blocks), dim3(64), 0, stream, dp, a);
  else
    hipLaunchKernelGGL((ddd::lean::integrate_kernel<kK, kD, 4>), dim3(blocks), dim3(64), 0, stream, dp, a);
}
bool launch_lean(const ddd_model* m, const ddd::IntegrateArgs& a, hipStream_t stream) {
  const ddd::DevParams& dp = m->dp;
  if (!ddd::lean::supports(dp)) return false;
  const int taps = dp.fixed ? 0 : dp.linear_taps, pairs = ddd::lean::column_pairs(dp.G);
  const int blocks = (a.batch + 64 / dp.N - 1) / (64 / dp.N);
  const int derivs = dp.D;
#define DDD_LEAN_CASE(TAPS, DERIVS) \
  if (taps == TAPS && derivs == DERIVS) {                                 \
    launch_lean_kd<TAPS, DERIVS>(pairs, dp, a, blocks, stream);           \
    return true;                                                          \
  }
  DDD_LEAN_CASE(0, 1) DDD_LEAN_CASE(0, 2) DDD_LEAN_CASE(0, 3) DDD_LEAN_CASE(0, 4)
  DDD_LEAN_CASE(3, 1) DDD_LEAN_CASE(3, 2) DDD_LEAN_CASE(3, 3)
  DDD_LEAN_CASE(5, 1) DDD_LEAN_CASE(5, 2) DDD_LEAN_CASE(5, 3)
  DDD_LEAN_CASE(7, 1) DDD_LEAN_CASE(7, 2)
#undef DDD_LEAN_CASE
  return false;   // (other tap counts: the MFMA-path kernels with the tower skipped)
}

template <typename ST>
int launch_integrate(ddd_model* m, ddd::IntegrateArgs a, hipStream_t stream) {
  if (a.batch == 0 || a.n_steps == 0) return DDD_OK;
  m->last_batch = a.batch;
  m->last_launch_streamed = false; m->last_launch_lean = false;
#ifdef DDD_PROBES   // profiling knobs (ddd_debug_set_option; profiles/r1_ablation.txt)
  a.prio_split = g_debug.prio_split;
  a.stagger = g_debug.stagger;
  a.trace = reinterpret_cast<unsigned long long*>(g_debug.trace_ptr);
  a.ablate = g_debug.ablate;
#endif
  m->last_launch_split = false;
  m->last_launch_quad = false; m->last_launch_half = false;
  if (m->kernel == DDD_KERNEL_MFMA && std::is_same<ST, float>::value && !m->explicit_kernel &&
      !g_debug.no_lean && launch_lean(m, a, stream)) {
    // fixed stencils / one-layer nets, float32 state, whole samples per wavefront: the
    // lane == grid point kernel (no matrix work to schedule around)
    m->last_launch_lean = true;
    DDD_HIP(hipGetLastError());
    return DDD_OK;
  }
  if (m->kernel == DDD_KERNEL_MFMA) {
    MfmaGeometry geo = mfma_geometry(m, a.batch);
    if (geo.rows == 64 && geo.wave_rows == 64 && std::is_same<ST, float>::value &&
        m->split_auto && !m->explicit_kernel && m->dp.w_final4_split != nullptr &&
        spec_equation(m, 64) >= 0) {
      // small ensembles: two 32-row wavefronts per sample (rhs_mfma.h kSplit)
      // measured (profiles/r4_ablation.txt, N = 64): B = 256: 1.63 x the one-wavefront kernel (twice
      // the SIMDs busy), B = 512: 0.97 x, B = 1024: 0.89 x (two coupled wavefronts per SIMD lose to one
      // free-running one) -- so only below three eighths of a wavefront per SIMD
      const int spg = 64 / m->dp.N;
      const int groups = (a.batch + spg - 1) / spg;
      if (8 * groups <= 3 * device_simds()) geo = {64, 32};
      // ... and FOUR 16-row wavefronts per group (kQuad: every layer on 16x16x4 MFMAs, one
      // group on the four SIMDs of a CU) while that leaves at most two wavefronts per SIMD
      // (measured, profiles/r6_ablation.txt)
      if (m->dp.w_quad != nullptr && 2 * groups <= device_simds()) geo = {64, 16};
    }
    // an explicit DDD_KERNEL_MFMA_ROWS64_W16 / _W32 where the model has no such kernel (float64
    // state, no per-equation specialisation): the one-wavefront geometry
    if (geo.rows == 64 && geo.wave_rows == 16 &&
        !(std::is_same<ST, float>::value && m->dp.w_quad != nullptr && spec_equation(m, 64) >= 0))
      geo = {64, 64};
    m->last_launch_split = geo.rows == 64 && geo.wave_rows == 32;
    m->last_launch_quad = geo.rows == 64 && geo.wave_rows == 16;
    if (geo.rows == 64 && geo.wave_rows == 64) launch_mfma_integrate<64, 64, ST>(m, a, stream);
    else if (geo.rows == 64 && geo.wave_rows == 16) launch_mfma_integrate<64, 16, ST>(m, a, stream);
    else if (geo.rows == 64) launch_mfma_integrate<64, 32, ST>(m, a, stream);
    else launch_mfma_integrate<256, 64, ST>(m, a, stream);
  } else if (use_weno_kernel(m)) {
    ddd::launch::weno_integrate(std::is_same<ST, double>::value, m->dp, a, stream);
  } else {
    int rc = check_generic_lds(m, (int)sizeof(ST));
    if (rc) return rc;
    const size_t lds = ddd::generic::lds_bytes(m->dp, (int)sizeof(ST));
    DDD_HIP(hipFuncSetAttribute(
        reinterpret_cast<const void*>(ddd::generic::integrate_kernel<ST>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((ddd::generic::integrate_kernel<ST>), dim3(a.batch),
                       dim3(ddd::generic::kThreads), lds, stream, m->dp, a);
  }
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int check_integrate_args(const ddd_model* m, int n_steps, int save_every,
                         const void* y0, const void* y_out, int batch) {
  int rc = check_batch(m, batch);
  if (rc) return rc;
  if (n_steps < 0) return fail(DDD_ERR_INVALID_ARGUMENT, "negative n_steps");
  if (save_every < 1) return fail(DDD_ERR_INVALID_ARGUMENT, "save_every must be >= 1");
  if (batch > 0 && n_steps > 0 && (y0 == nullptr || y_out == nullptr))
    return fail(DDD_ERR_INVALID_ARGUMENT, "y0 / y_out is NULL");
  return DDD_OK;
}

// Large ensembles on the per-equation MFMA kernels are advanced as TWO
// half-ensembles (contiguous sample slabs, independent of each other) on two
// internal streams, each launch sized to half the machine: while one half is at
// a kernel boundary (drain, dispatch, 29 KB of weights per wavefront before the
// first MFMA) the other half's wavefronts have the matrix pipes to themselves,
// which a single wavefront per SIMD nearly saturates.  Still one fused launch per
// substep for every sample; results are bit-identical (same kernel, same
// arithmetic).  Used by ddd_integrate_fixed (per-substep / per-step modes) and,
// across calls, by ddd_rk_substep / ddd_time_derivative inside a
// ddd_stream_fork .. ddd_stream_join region.
constexpr int kMaxParts = 4;
struct SlabPlan {
  int halves = 1;                                 // slabs advanced side by side
  int half_batch[kMaxParts] = {0, 0, 0, 0};
  int slab_first[kMaxParts] = {0, 0, 0, 0};       // first sample of each slab
};

SlabPlan plan_slabs(const ddd_model* m, int batch) {
  SlabPlan plan;
  plan.half_batch[0] = batch;
  if (m->kernel != DDD_KERNEL_MFMA || m->explicit_kernel || g_debug.no_spec) return plan;
  const MfmaGeometry geo = mfma_geometry(m, batch);
  const int spg = geo.rows / m->dp.N;
  const int groups = (batch + spg - 1) / spg;
  const int capacity = geo.rows == 64 ? 2 * device_simds() : device_simds() / 2;
  if (geo.wave_rows != 64 || spec_equation(m, geo.rows) < 0 || groups < 2 * capacity) return plan;
  plan.halves = 2;
  if (g_debug.substep_parts > 0) plan.halves = std::min(g_debug.substep_parts, kMaxParts);
  const int per = ((groups + plan.halves - 1) / plan.halves) * spg;   // whole workgroups
  int first = 0;
  for (int i = 0; i < plan.halves; ++i) {
    plan.slab_first[i] = first;
    plan.half_batch[i] = std::max(0, std::min(per, batch - first));
    first += plan.half_batch[i];
  }
  return plan;
}

// The internal streams start after everything already enqueued on `stream` ...
int fork_lanes(ddd_model* m, hipStream_t stream, int halves) {
  for (int i = 0; i < halves; ++i) {
    if (m->aux_stream[i] == nullptr)
      DDD_HIP(hipStreamCreateWithFlags(&m->aux_stream[i], hipStreamNonBlocking));
    if (m->ev_join[i] == nullptr)
      DDD_HIP(hipEventCreateWithFlags(&m->ev_join[i], hipEventDisableTiming));
  }
  if (m->ev_fork == nullptr)
    DDD_HIP(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
  DDD_HIP(hipEventRecord(m->ev_fork, stream));
  for (int i = 0; i < halves; ++i) DDD_HIP(hipStreamWaitEvent(m->aux_stream[i], m->ev_fork, 0));
  // (No start offset between the chains: whatever it is, they settle into the
  // same steady state within a few launches -- measured 0 .. 1.75 half-launches,
  // 72.4 +- 0.1 % throughout, profiles/r3_ablation.txt.)
  return DDD_OK;
}

// ... and `stream` continues after everything enqueued on them.  Never skipped: an
// error inside a forked region still joins before it is reported.
int join_lanes(ddd_model* m, hipStream_t stream, int halves) {
  for (int i = 0; i < halves; ++i) {
    DDD_HIP(hipEventRecord(m->ev_join[i], m->aux_stream[i]));
    DDD_HIP(hipStreamWaitEvent(stream, m->ev_join[i], 0));
  }
  return DDD_OK;
}

// Entry points other than ddd_rk_substep / ddd_time_derivative end a caller's
// chained region (ddd_stream_fork) before they touch the model: its stream then
// sees every substep enqueued so far.
int chain_close(ddd_model* m) {
  int rc = DDD_OK;
  if (m->ring != nullptr) {
    std::lock_guard<std::mutex> lk(m->ring->mu);
    if (!ring_stop_locked(m->ring))
      rc = fail(DDD_ERR_HIP, "command ring: the persistent kernel made no progress for 10 s");
  }
  if (m->chain.open && m->chain.forked && m->chain.halves > 1)
    rc = join_lanes(m, m->chain.stream, m->chain.halves);
  m->chain.open = false;
  m->chain.forked = false;
  return rc;
}

// The region can run on the command ring: a per-equation kernel on one-wave groups (N | 64),
// the whole ensemble from sample 0, no derivative / coefficient views.
bool ring_eligible(const ddd_model* m, const ddd::SubstepArgs& a) {
  if (m->region_mode != DDD_REGION_RING) return false;   // (opt-in: measured equal to the launches)
  if (m->kernel != DDD_KERNEL_MFMA || m->explicit_kernel || g_debug.no_spec) return false;
  if (a.batch <= 0 || a.derivs_out != nullptr || a.coeffs_out != nullptr) return false;
  if (use_stream_kernel(m, a)) return false;
  MfmaGeometry geo = mfma_geometry(m, a.batch);
  if (geo.wave_rows == 16) geo = {64, 64};
  return geo.rows == 64 && geo.wave_rows == 64 && spec_equation(m, 64) >= 0;
}

int ring_create(ddd_model* m) {
  Ring* rg = new Ring();
  const size_t slot_bytes = (size_t)ddd::kRingSlots * ddd::kRingSlotChunks * 16;
  hipError_t err = hipHostMalloc(&rg->slots, slot_bytes, hipHostMallocCoherent | hipHostMallocMapped);
  if (err == hipSuccess)
    err = hipHostMalloc(reinterpret_cast<void**>(&rg->done), (ddd::kRingSlots + 32 + 2048) * sizeof(unsigned),
                        hipHostMallocCoherent | hipHostMallocMapped);
  if (err == hipSuccess)
    err = hipMalloc(reinterpret_cast<void**>(&rg->d_count), (ddd::kRingSlots + 16) * sizeof(unsigned));
  if (err == hipSuccess)   // fine-grained: every XCD's L2 sees the relaying wavefront's stores
    err = hipExtMallocWithFlags(&rg->d_slots, slot_bytes, hipDeviceMallocFinegrained);
  if (err == hipSuccess)   // (every slot's last group resets its count; the relay lock starts open)
    err = hipMemset(rg->d_count, 0, (ddd::kRingSlots + 16) * sizeof(unsigned));
  if (err == hipSuccess) err = hipMemset(rg->d_slots, 0, slot_bytes);
  if (err == hipSuccess) err = hipDeviceSynchronize();
  if (err != hipSuccess) {
    if (rg->slots) (void)hipHostFree(rg->slots);
    if (rg->done) (void)hipHostFree(rg->done);
    if (rg->d_count) (void)hipFree(rg->d_count);
    if (rg->d_slots) (void)hipFree(rg->d_slots);
    delete rg;
    return fail(DDD_ERR_HIP, "command ring: %s", hipGetErrorString(err));
  }
  std::memset(rg->slots, 0, slot_bytes);
  std::memset(rg->done, 0, (ddd::kRingSlots + 32 + 2048) * sizeof(unsigned));   // (+ status, trace stamps)
  rg->status = rg->done + ddd::kRingSlots;
  rg->parker = std::thread(ring_parker, rg);
  m->ring = rg;
  return DDD_OK;
}

void ring_destroy(ddd_model* m) {
  Ring* rg = m->ring;
  if (rg == nullptr) return;
  {
    std::lock_guard<std::mutex> lk(rg->mu);
    (void)ring_stop_locked(rg);
    rg->quit = true;
  }
  rg->cv.notify_all();
  rg->parker.join();
  // the kernel reads the ring until it ends (a stream the caller has already destroyed: the device)
  if (rg->stream == nullptr || hipStreamSynchronize(rg->stream) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipDeviceSynchronize();
  }
  (void)hipHostFree(rg->slots);
  (void)hipHostFree(rg->done);
  (void)hipFree(rg->d_count);
  (void)hipFree(rg->d_slots);
  delete rg;
  m->ring = nullptr;
}

// One command = one ddd_rk_substep call.  Starts the persistent kernel when none is waiting.
int ring_post(ddd_model* m, const ddd::SubstepArgs& a, hipStream_t stream) {
  if (m->ring == nullptr) { int rc = ring_create(m); if (rc) return rc; }
  Ring* rg = m->ring;
  std::unique_lock<std::mutex> lk(rg->mu);
  if (rg->dead || *reinterpret_cast<const volatile unsigned*>(rg->status) != 0u) {
    rg->dead = true; rg->running = false;
    return fail(DDD_ERR_HIP, "command ring: the persistent kernel gave up waiting (watchdog) or "
                             "made no progress; the region's results are undefined");
  }
  if (rg->running && rg->stream != stream) (void)ring_stop_locked(rg);
  if (!rg->running) {
    m->dp.dpp_rol = dpp_wave_rol_ok();
    const int spg = 64 / m->dp.N;
    const int groups = (a.batch + spg - 1) / spg;
    rg->grid = std::min(groups, 2 * device_simds());
    ddd::RingArgs r{};
    r.slots = rg->slots; r.dev_slots = rg->d_slots; r.count = rg->d_count; r.done = rg->done; r.status = rg->status;
    r.first_index = (unsigned)rg->next_index;
    r.watchdog_ticks = rg->watchdog_ms * 100000u;   // 100 MHz
    const int eq = spec_equation(m, 64);
#define DDD_RING_CASE(EQ) \
    case EQ: ddd::launch::substep_ring_spec<EQ>(m->dp, r, rg->grid, stream); break;
    switch (eq) {
      DDD_RING_CASE(ddd::EQ_BURGERS)
      DDD_RING_CASE(ddd::EQ_BURGERS_CONS)
      DDD_RING_CASE(ddd::EQ_KDV)
      DDD_RING_CASE(ddd::EQ_KDV_CONS)
      DDD_RING_CASE(ddd::EQ_KS)
      DDD_RING_CASE(ddd::EQ_KS_CONS)
      default: return fail(DDD_ERR_UNSUPPORTED, "command ring: no per-equation kernel");
    }
#undef DDD_RING_CASE
    DDD_HIP(hipGetLastError());
    rg->running = true;
    rg->stream = stream;
    rg->launch_index = rg->next_index;
    ++rg->launches;
    rg->last_post = std::chrono::steady_clock::now();
    rg->cv.notify_all();
  }
  if (!ring_wait_room(rg, rg->next_index)) {
    rg->dead = true;
    (void)ring_stop_locked(rg);
    return fail(DDD_ERR_HIP, "command ring: the persistent kernel made no progress for 10 s");
  }
  unsigned payload[3 * ddd::kRingChunks] = {};
  const auto put64 = [&](int j, unsigned long long v) {
    payload[j] = (unsigned)v; payload[j + 1] = (unsigned)(v >> 32);
  };
  unsigned long long tbits; std::memcpy(&tbits, &a.t, 8);
  put64(0, tbits);
  put64(2, (unsigned long long)(uintptr_t)a.y_in);
  put64(4, (unsigned long long)(uintptr_t)a.y_base);
  put64(6, (unsigned long long)(uintptr_t)a.y_out);
  put64(8, (unsigned long long)(uintptr_t)a.acc_in);
  put64(10, (unsigned long long)(uintptr_t)a.acc_out);
  std::memcpy(&payload[12], &a.c1, 4);
  std::memcpy(&payload[13], &a.c2, 4);
  payload[14] = (unsigned)a.batch;
  ring_write(rg, rg->next_index, payload);
  ++rg->next_index;
  ++rg->commands;
  rg->last_post = std::chrono::steady_clock::now();
  m->last_batch = a.batch;
  m->last_launch_streamed = false; m->last_launch_lean = false;
  m->last_launch_split = false; m->last_launch_quad = false; m->last_launch_half = false;
  return DDD_OK;
}

// One substep of a caller-owned Runge-Kutta loop.  Outside a chained region: one
// launch on the caller's stream.  Inside (same stream): slab i of this call is
// ordered after slab i of the previous call only -- the two half-ensemble chains
// of ddd_integrate_fixed, kept alive across calls.
int substep_entry(ddd_model* m, const ddd::SubstepArgs& a, hipStream_t stream) {
  ddd_model::Chain& ch = m->chain;
  if (!ch.open || stream != ch.stream || a.derivs_out != nullptr || a.coeffs_out != nullptr)
    return launch_substep(m, a, stream);
  if (ring_eligible(m, a)) {
    if (ch.forked && ch.halves > 1) {   // (a region that started on the two chains)
      int rc = join_lanes(m, stream, ch.halves);
      ch.forked = false;
      if (rc) return rc;
    }
    return ring_post(m, a, stream);
  }
  if (m->ring != nullptr) {   // this call takes launches: behind every command posted so far
    std::lock_guard<std::mutex> lk(m->ring->mu);
    (void)ring_stop_locked(m->ring);
  }
  if (!ch.forked || ch.batch != a.batch) {
    if (ch.forked && ch.halves > 1) {   // another batch: other slabs, so join first
      int rc = join_lanes(m, stream, ch.halves);
      if (rc) return rc;
    }
    const SlabPlan plan = plan_slabs(m, a.batch);
    ch.batch = a.batch;
    ch.halves = plan.halves;
    for (int i = 0; i < kMaxParts; ++i) {
      ch.half_batch[i] = plan.half_batch[i];
      ch.slab_first[i] = plan.slab_first[i];
    }
    ch.forked = true;
    if (ch.halves > 1) {
      int rc = fork_lanes(m, stream, ch.halves);
      if (rc) { ch.forked = false; return rc; }
    }
  }
  if (ch.halves <= 1) return launch_substep(m, a, stream);
  int rc = DDD_OK;
  for (int hf = 0; hf < ch.halves && rc == DDD_OK; ++hf) {
    if (ch.half_batch[hf] == 0) continue;
    const size_t off = (size_t)ch.slab_first[hf] * m->dp.N;
    ddd::SubstepArgs h = a;
    h.batch = ch.half_batch[hf];
    h.y_in = a.y_in + off;
    if (a.y_base != nullptr) h.y_base = a.y_base + off;
    if (a.y_out != nullptr) h.y_out = a.y_out + off;
    if (a.acc_in != nullptr) h.acc_in = a.acc_in + off;
    if (a.acc_out != nullptr) h.acc_out = a.acc_out + off;
    rc = launch_substep(m, h, m->aux_stream[hf], ch.slab_first[hf], ch.halves);
  }
  m->last_batch = a.batch;
  if (rc) {   // do not leave the internal streams running ahead of a failed call
    (void)join_lanes(m, stream, ch.halves);
    ch.forked = false;
  }
  return rc;
}

}  // namespace

extern "C" {

int ddd_abi_version(void) { return DDD_ABI_VERSION; }

const char* ddd_last_error(void) { return g_error.c_str(); }

int ddd_scheme_stages(int scheme) {
  ddd::Tableau tab;
  if (make_tableau(scheme, &tab) != DDD_OK) return -1;
  return tab.stages;
}

int ddd_model_create(const ddd_config* cfg, const float* weights, size_t n_weights,
                     const float* nullspace, size_t n_nullspace, const float* bias,
                     size_t n_bias, ddd_model** out) {
  if (out == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  int rc = common_config_checks(cfg);
  if (rc) return rc;
  if (cfg->num_layers < 1 || cfg->num_layers > DDD_MAX_LAYERS)
    return fail(DDD_ERR_INVALID_ARGUMENT,
                "num_layers = %d out of range [1, %d] (fold num_layers = 0 models "
                "into ddd_baseline_create)", cfg->num_layers, DDD_MAX_LAYERS);
  if (cfg->filter_size < 1 || cfg->kernel_size < 1)
    return fail(DDD_ERR_INVALID_ARGUMENT, "filter_size / kernel_size must be >= 1");
  if (cfg->activation < DDD_ACT_RELU || cfg->activation > DDD_ACT_ELU)
    return fail(DDD_ERR_INVALID_ARGUMENT, "unknown activation %d", cfg->activation);
  if (cfg->model_target < DDD_TARGET_COEFFICIENTS || cfg->model_target > DDD_TARGET_FLUX)
    return fail(DDD_ERR_INVALID_ARGUMENT, "unknown model_target %d", cfg->model_target);
  if (!(cfg->standard_deviation > 0.0))
    return fail(DDD_ERR_INVALID_ARGUMENT, "standard_deviation must be positive");
  if (weights == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "weights is NULL");

  ddd_model* m = new ddd_model();
  m->cfg = *cfg;
  fill_equation(*cfg, &m->dp);
  ddd::DevParams& dp = m->dp;
  dp.fixed = 0;
  dp.target = cfg->model_target;
  dp.L = cfg->num_layers;
  dp.F = cfg->filter_size;
  dp.K = cfg->kernel_size;
  dp.act = cfg->activation;
  dp.pao = cfg->polynomial_accuracy_order;
  dp.unbiased = cfg->ensure_unbiased_coefficients;

  const bool projected = dp.target == ddd::TARGET_COEFFICIENTS && dp.pao > 0;
  int c_out = 1;
  if (dp.target == ddd::TARGET_COEFFICIENTS) {
    if (projected) {
      c_out = 0;
      for (int d = 0; d < dp.D; ++d) {
        if (cfg->input_sizes[d] < 1 || cfg->input_sizes[d] > dp.G) {
          delete m;
          return fail(DDD_ERR_INVALID_ARGUMENT, "input_sizes[%d] = %d out of range [1, G]",
                      d, cfg->input_sizes[d]);
        }
        dp.in_start[d] = c_out;
        dp.in_size[d] = cfg->input_sizes[d];
        dp.ns_off[d] = c_out * dp.G;
        c_out += cfg->input_sizes[d];
      }
    } else {
      c_out = dp.D * dp.G;
    }
  } else if (dp.target == ddd::TARGET_SPACE_DERIVATIVES) {
    c_out = dp.D;
  }
  dp.C_out = c_out;

  size_t off = 0;
  int64_t fma = 0;
  for (int l = 0; l < dp.L; ++l) {
    dp.cin[l] = l == 0 ? 1 : dp.F;
    dp.cout[l] = l == dp.L - 1 ? c_out : dp.F;
    dp.w_off[l] = (int)off;
    off += (size_t)dp.K * dp.cin[l] * dp.cout[l];
    dp.b_off[l] = (int)off;
    off += (size_t)dp.cout[l];
    fma += (int64_t)dp.K * dp.cin[l] * dp.cout[l];
  }
  if (n_weights != off) {
    delete m;
    return fail(DDD_ERR_INVALID_ARGUMENT,
                "n_weights = %zu, configuration needs %zu floats", n_weights, off);
  }
  if (dp.target == ddd::TARGET_COEFFICIENTS) {
    if (projected) for (int d = 0; d < dp.D; ++d) fma += (int64_t)dp.in_size[d] * dp.G;
    fma += (int64_t)dp.D * dp.G;
  }
  m->fma_per_point = fma;

  if (projected) {
    if (nullspace == nullptr || bias == nullptr || n_nullspace != (size_t)c_out * dp.G ||
        n_bias != (size_t)dp.D * dp.G) {
      delete m;
      return fail(DDD_ERR_INVALID_ARGUMENT,
                  "nullspace/bias required: expected %zu and %zu floats, got %zu and %zu",
                  (size_t)c_out * dp.G, (size_t)dp.D * dp.G, n_nullspace, n_bias);
    }
  }

  std::vector<float> wv(weights, weights + n_weights);
  rc = upload(wv, &m->d_weights);
  dp.weights = m->d_weights;
  if (!rc) {
    // the generic kernel stages a layer's weights in LDS with the output channels
    // padded to fours ([K][cin][c4] + bias [c4]): kept in that form, so that the
    // staging is a straight float4 copy (rhs_generic.h)
    std::vector<float> w4;
    for (int l = 0; l < dp.L; ++l) {
      const int c4 = (dp.cout[l] + 3) & ~3;
      dp.w4_off[l] = (int)w4.size();
      const float* w = wv.data() + dp.w_off[l];
      const float* b = wv.data() + dp.b_off[l];
      for (int kc = 0; kc < dp.K * dp.cin[l]; ++kc)
        for (int col = 0; col < c4; ++col)
          w4.push_back(col < dp.cout[l] ? w[(size_t)kc * dp.cout[l] + col] : 0.0f);
      for (int col = 0; col < c4; ++col) w4.push_back(col < dp.cout[l] ? b[col] : 0.0f);
    }
    rc = upload(w4, &m->d_weights4);
    dp.weights4 = m->d_weights4;
  }
  if (!rc && projected) {
    std::vector<float> nv(nullspace, nullspace + n_nullspace);
    std::vector<float> bv(bias, bias + n_bias);
    rc = upload(nv, &m->d_nullspace);
    if (!rc) rc = upload(bv, &m->d_bias);
    dp.nullspace = m->d_nullspace;
    dp.bias = m->d_bias;
  }
  if (!rc) {
    decide_mfma(m);
    if (m->mfma_ok && linear_eligible(dp)) {
      // coeff[d][g] = B[d][g] + sum_k M[k][d][g] (u / std)[x + k - K/2]: the conv layer, the
      // null-space projection and the accuracy bias folded in float64, rounded once
      std::memset(dp.ns8, 0, sizeof(dp.ns8));
      std::memset(dp.bias8, 0, sizeof(dp.bias8));
      const float* w = wv.data() + dp.w_off[0];   // [K][1][C_out]
      const float* b = wv.data() + dp.b_off[0];
      for (int d = 0; d < dp.D; ++d)
        for (int g = 0; g < dp.G; ++g) {
          if (projected) {
            double acc = (double)bias[d * dp.G + g];
            for (int j = 0; j < dp.in_size[d]; ++j)
              acc += (double)b[dp.in_start[d] + j] * (double)nullspace[dp.ns_off[d] + j * dp.G + g];
            dp.bias8[d][g] = (float)acc;
            for (int k = 0; k < dp.K; ++k) {
              double mk = 0.0;
              for (int j = 0; j < dp.in_size[d]; ++j)
                mk += (double)w[(size_t)k * dp.C_out + dp.in_start[d] + j] *
                      (double)nullspace[dp.ns_off[d] + j * dp.G + g];
              dp.ns8[k * dp.D + d][g] = (float)mk;
            }
          } else {   // polynomial_accuracy_order 0: the layer emits the coefficients themselves
            dp.bias8[d][g] = b[d * dp.G + g];
            for (int k = 0; k < dp.K; ++k)
              dp.ns8[k * dp.D + d][g] = w[(size_t)k * dp.C_out + d * dp.G + g];
          }
        }
      dp.linear_taps = dp.K;
      dp.folded = 0;
      dp.dsel_bits = 0;
      dp.dsel_valid = 0;
    } else if (m->mfma_ok) {
      const bool fold_shape = dp.D <= 2 && dp.G >= 6 && dp.G <= ddd::kGMax && !m->wide;
      rc = upload_padded_tables(m, projected ? nullspace : nullptr, projected ? bias : nullptr,
                                dp.target == ddd::TARGET_COEFFICIENTS && !projected &&
                                    !fold_shape && !m->wide);
      if (!rc) {
        // (the generic kernel keeps the model's own K, F and weights: a model that
        // falls back to it -- N > 256, ddd_set_kernel(generic) -- never runs the
        // padded net; only the MFMA packing sees the embedding)
        std::vector<float> padded;
        NetLayout net;
        net.weights = wv.data();
        for (int l = 0; l < dp.L; ++l) { net.w_off[l] = dp.w_off[l]; net.b_off[l] = dp.b_off[l]; }
        if (dp.K != m->tower_k || dp.F != 32 * m->tower_cb)
          embed_tower(dp, wv, m->tower_k, 32 * m->tower_cb, &padded, &net);
        rc = pack_mfma_weights(m, net);
      }
    }
  }
  if (rc) { ddd_model_destroy(m); return rc; }
  *out = m;
  return DDD_OK;
}

int ddd_baseline_create(const ddd_config* cfg, const float* stencils, size_t n_stencils,
                        ddd_model** out) {
  if (out == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  int rc = common_config_checks(cfg);
  if (rc) return rc;
  if (stencils == nullptr ||
      n_stencils != (size_t)cfg->num_derivatives * cfg->stencil_size)
    return fail(DDD_ERR_INVALID_ARGUMENT, "stencils must hold D*G = %d floats, got %zu",
                cfg->num_derivatives * cfg->stencil_size, n_stencils);
  ddd_model* m = new ddd_model();
  m->cfg = *cfg;
  fill_equation(*cfg, &m->dp);
  m->dp.fixed = 1;
  m->dp.weno = cfg->weno_reconstruction != 0;
  if (m->dp.weno && (cfg->num_derivatives < 2 || cfg->equation < DDD_EQ_BURGERS_GODUNOV)) {
    delete m;
    return fail(DDD_ERR_INVALID_ARGUMENT,
                "weno_reconstruction needs a Godunov-flux equation (u_minus, u_plus)");
  }
  m->dp.stddev = 1.0f;
  m->dp.inv_stddev = 1.0f;
  m->dp.exact_div = 0;
  m->fma_per_point = (int64_t)cfg->num_derivatives * cfg->stencil_size;
  std::vector<float> sv(stencils, stencils + n_stencils);
  rc = upload(sv, &m->d_bias);
  m->dp.bias = m->d_bias;
  if (!rc) {
    decide_mfma(m);
    // the padded stencil table also feeds the streaming kernel (any N <= 1024)
    if (m->mfma_ok || m->dp.G <= ddd::kGWide) rc = upload_padded_tables(m, nullptr, stencils);
  }
  if (rc) { ddd_model_destroy(m); return rc; }
  *out = m;
  return DDD_OK;
}

int ddd_spectral_create(const ddd_config* cfg, const double* kernels, size_t n_kernels,
                        ddd_model** out) {
  if (out == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  int rc = common_config_checks(cfg);
  if (rc) return rc;
  if (cfg->equation != DDD_EQ_BURGERS && cfg->equation != DDD_EQ_KDV &&
      cfg->equation != DDD_EQ_KS)
    return fail(DDD_ERR_UNSUPPORTED,
                "spectral models exist for the non-flux equations only (integrate.py:346)");
  if (cfg->num_points > ddd::spectral::kMaxPoints || cfg->num_points % 2)
    return fail(DDD_ERR_UNSUPPORTED, "spectral models need an even num_points <= %d (got %d)",
                ddd::spectral::kMaxPoints, cfg->num_points);
  if (kernels == nullptr || n_kernels != (size_t)cfg->num_derivatives * cfg->num_points)
    return fail(DDD_ERR_INVALID_ARGUMENT, "kernels must hold D*N = %d doubles, got %zu",
                cfg->num_derivatives * cfg->num_points, n_kernels);
  ddd_model* m = new ddd_model();
  m->cfg = *cfg;
  fill_equation(*cfg, &m->dp);
  m->spectral = true;
  m->kernel = DDD_KERNEL_GENERIC;
  m->mfma_reason = "spectral model";
  m->fma_per_point = (int64_t)cfg->num_derivatives * cfg->num_points;
  m->sp.equation = cfg->equation;
  m->sp.N = cfg->num_points;
  m->sp.D = cfg->num_derivatives;
  m->sp.eta = cfg->eta;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_kernels), n_kernels * sizeof(double));
  if (e == hipSuccess)
    e = hipMemcpy(m->d_kernels, kernels, n_kernels * sizeof(double), hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    ddd_model_destroy(m);
    return fail(DDD_ERR_HIP, "spectral kernels upload: %s", hipGetErrorString(e));
  }
  m->sp.kernels = m->d_kernels;
  // FFT mode for the large exact grids (N a power of two >= 512; rhs_spectral.h): the same
  // D circulant operators as diagonal multipliers in Fourier space, mult[d] = DFT(kernel[d]).
  // Formed HERE from the caller's kernels, so every Nyquist / sign convention of the call
  // that produced them carries over unchanged; a direct O(N^2) DFT in long double (once per
  // model: 12 M complex terms at N = 2048, D = 3), twiddles exp(-2 pi i m / N) likewise.
  const int n = cfg->num_points;
  if (n >= ddd::spectral::kFftMinPoints && (n & (n - 1)) == 0 && !g_debug.no_fft) {
    const long double two_pi = 6.283185307179586476925286766559005768L;
    std::vector<long double> cs(n), sn(n);
    for (int j = 0; j < n; ++j) {
      cs[j] = cosl(two_pi * (long double)j / (long double)n);
      sn[j] = sinl(two_pi * (long double)j / (long double)n);
    }
    std::vector<double2> mult((size_t)cfg->num_derivatives * n), tw(n / 2);
    for (int d = 0; d < cfg->num_derivatives; ++d)
      for (int k = 0; k < n; ++k) {
        long double re = 0.0L, im = 0.0L;
        for (int j = 0; j < n; ++j) {
          const int a = (int)(((long long)j * k) % n);
          const long double c = (long double)kernels[(size_t)d * n + j];
          re += c * cs[a];
          im -= c * sn[a];
        }
        mult[(size_t)d * n + k] = make_double2((double)re, (double)im);
      }
    for (int i = 0; i < n / 2; ++i) tw[i] = make_double2((double)cs[i], (double)-sn[i]);
    int rc2 = upload(mult, &m->d_fft_mult);
    if (rc2 == DDD_OK) rc2 = upload(tw, &m->d_fft_twiddle);
    if (rc2 != DDD_OK) { ddd_model_destroy(m); return rc2; }
    m->sp.fft_mult = m->d_fft_mult;
    m->sp.fft_twiddle = m->d_fft_twiddle;
    m->sp.fft_log2n = 31 - __builtin_clz((unsigned)n);
    m->fma_per_point = (int64_t)(2 + (cfg->num_derivatives + 1) / 2) * 5 * m->sp.fft_log2n / 2;
  }
  *out = m;
  return DDD_OK;
}

namespace {
int launch_spectral(ddd_model* m, const ddd::spectral::SubstepArgs64& a, hipStream_t stream) {
  if (a.batch == 0) return DDD_OK;
  const size_t lds = ddd::spectral::lds_bytes(m->sp);
  DDD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ddd::spectral::substep_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(ddd::spectral::substep_kernel, dim3(a.batch),
                     dim3(ddd::spectral::kThreads), lds, stream, m->sp, a);
  DDD_HIP(hipGetLastError());
  m->last_batch = a.batch;
  return DDD_OK;
}
}  // namespace

int ddd_time_derivative_f64(ddd_model* m, double t, const double* y, double* dydt, int batch,
                            void* stream) {
  (void)t;   // the equations of motion are autonomous; forcing stays on the host
  int rc = check_batch(m, batch, true);
  if (rc) return rc;
  if (batch > 0 && (!y || !dydt)) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  ddd::spectral::SubstepArgs64 a{};
  a.y_in = y; a.c1 = 1.0; a.y_out = dydt; a.batch = batch;
  return launch_spectral(m, a, static_cast<hipStream_t>(stream));
}

int ddd_rk_substep_f64(ddd_model* m, double t, const double* y_in, const double* y_base,
                       double c1, double* y_out, const double* acc_in, double c2,
                       double* acc_out, int batch, void* stream) {
  (void)t;
  int rc = check_batch(m, batch, true);
  if (rc) return rc;
  if (batch > 0 && !y_in) return fail(DDD_ERR_INVALID_ARGUMENT, "y_in is NULL");
  if (batch > 0 && !y_out && !acc_out)
    return fail(DDD_ERR_INVALID_ARGUMENT, "both y_out and acc_out are NULL");
  ddd::spectral::SubstepArgs64 a{};
  a.y_in = y_in; a.y_base = y_base; a.c1 = c1; a.y_out = y_out;
  a.acc_in = acc_in; a.c2 = c2; a.acc_out = acc_out; a.batch = batch;
  return launch_spectral(m, a, static_cast<hipStream_t>(stream));
}

int ddd_model_destroy(ddd_model* m) {
  if (m == nullptr) return DDD_OK;
  (void)chain_close(m);   // the caller's stream sees every substep before the buffers go
  ring_destroy(m);
  free_dev(m->d_weights); free_dev(m->d_weights4); free_dev(m->d_nullspace); free_dev(m->d_bias);
  free_dev(m->d_w_hidden);
  free_dev(m->d_w_input);
  free_dev(m->d_w_hidden_half); free_dev(m->d_w_final4_half); free_dev(m->d_w_t16);
  free_dev(m->d_w_final4); free_dev(m->d_w_final4_rt); free_dev(m->d_w_final4_split); free_dev(m->d_w_quad); free_dev(m->d_frc); free_dev(m->d_sp); free_dev(m->d_trig);
  if (m->d_runs != nullptr) (void)hipFree(m->d_runs);
  free_dev(m->d_scratch);
  for (auto& slot : m->time_slot) {
    if (slot.done != nullptr) {
      if (slot.in_flight) (void)hipEventSynchronize(slot.done);
      (void)hipEventDestroy(slot.done);
    }
    if (slot.host != nullptr) (void)hipHostFree(slot.host);
    free_dev(slot.dev);
  }
  for (int i = 0; i < 4; ++i) {
    if (m->aux_stream[i] != nullptr) (void)hipStreamDestroy(m->aux_stream[i]);
    if (m->ev_join[i] != nullptr) (void)hipEventDestroy(m->ev_join[i]);
  }
  if (m->ev_fork != nullptr) (void)hipEventDestroy(m->ev_fork);
  if (m->d_kernels != nullptr) (void)hipFree(m->d_kernels);
  if (m->d_fft_mult != nullptr) (void)hipFree(m->d_fft_mult);
  if (m->d_fft_twiddle != nullptr) (void)hipFree(m->d_fft_twiddle);
  if (m->d_scratch64 != nullptr) (void)hipFree(m->d_scratch64);
  delete m;
  return DDD_OK;
}

int ddd_clear_forcing(ddd_model* m) {
  if (m != nullptr) { int rcj = chain_close(m); if (rcj) return rcj; }
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  free_dev(m->d_frc); free_dev(m->d_sp); free_dev(m->d_trig);
  if (m->d_runs != nullptr) (void)hipFree(m->d_runs);
  m->d_frc = nullptr; m->d_sp = nullptr; m->d_trig = nullptr; m->d_runs = nullptr;
  m->dp.trig = nullptr; m->dp.runs = nullptr;
  m->dp.forced = 0; m->dp.P = 0; m->dp.n_k = 0; m->dp.forcing_batch = 0;
  m->dp.inv_P = 0.0f; m->dp.inv_nk = 1.0f;
  m->dp.frc = nullptr; m->dp.sp = nullptr;
  return DDD_OK;
}

int ddd_set_forcing(ddd_model* m, int batch, int nparams, const float* amplitude,
                    const float* omega, const float* phase, const int32_t* k_index,
                    const float* spatial_phase, int n_k) {
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  if (m->spectral)
    return fail(DDD_ERR_UNSUPPORTED,
                "spectral models carry no forcing: apply finalize_time_derivative on the host");
  if (batch < 1 || nparams < 1 || n_k < 1)
    return fail(DDD_ERR_INVALID_ARGUMENT, "batch, nparams and n_k must be >= 1");
  if (!amplitude || !omega || !phase || !k_index || !spatial_phase)
    return fail(DDD_ERR_INVALID_ARGUMENT, "NULL forcing table");
  int rc = ddd_clear_forcing(m);
  if (rc) return rc;
  if (!is_forced_family(m->cfg.equation)) return DDD_OK;   // finalize is the identity
  const size_t count = (size_t)batch * nparams;
  for (size_t i = 0; i < count; ++i)
    if (k_index[i] < 0 || k_index[i] >= n_k)
      return fail(DDD_ERR_INVALID_ARGUMENT, "k_index[%zu] = %d outside [0, %d)", i,
                  k_index[i], n_k);
  // Each sample's modes are stored sorted by k_index (stable), so that the
  // kernels sum the modes sharing a wavenumber over a contiguous range.
  std::vector<float4> packed(count);
  std::vector<int> order(nparams);
  for (int b = 0; b < batch; ++b) {
    const size_t base = (size_t)b * nparams;
    for (int j = 0; j < nparams; ++j) order[j] = j;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) {
      return k_index[base + x] < k_index[base + y];
    });
    for (int j = 0; j < nparams; ++j) {
      const size_t i = base + order[j];
      float kbits;
      const int32_t ki = k_index[i];
      std::memcpy(&kbits, &ki, sizeof(kbits));
      packed[base + j] = make_float4(amplitude[i], omega[i], phase[i], kbits);
    }
  }
  std::vector<float> sp(spatial_phase, spatial_phase + (size_t)n_k * m->dp.N);
  // cos / sin of the (float32) spatial phase per grid point, [N][12] zero
  // padded (entry 2 k = cos, 2 k + 1 = sin of phase table row k; n_k <= 6)
  std::vector<float> trig((size_t)m->dp.N * 12, 0.0f);
  for (int x = 0; x < m->dp.N; ++x)
    for (int k = 0; k < n_k && k < 6; ++k) {
      const double theta = (double)sp[(size_t)k * m->dp.N + x];
      trig[(size_t)x * 12 + 2 * k + 0] = (float)std::cos(theta);
      trig[(size_t)x * 12 + 2 * k + 1] = (float)std::sin(theta);
    }
  // runs[b][kk] = first (sorted) mode of sample b whose k index is >= kk: the
  // kernels' per-(sample, k) sums cover modes [runs[kk], runs[kk + 1])
  std::vector<unsigned char> runs((size_t)batch * 8, 0);
  if (nparams < 256)
    for (int b = 0; b < batch; ++b) {
      int mode = 0;
      for (int kk = 0; kk < 8; ++kk) {
        while (mode < nparams) {
          int32_t ki;
          std::memcpy(&ki, &packed[(size_t)b * nparams + mode].w, sizeof(ki));
          if (ki >= kk) break;
          ++mode;
        }
        runs[(size_t)b * 8 + kk] = (unsigned char)mode;
      }
    }
  rc = upload(packed, &m->d_frc);
  if (!rc) rc = upload(sp, &m->d_sp);
  if (!rc) rc = upload(trig, &m->d_trig);
  if (!rc) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&m->d_runs), runs.size());
    if (e == hipSuccess)
      e = hipMemcpy(m->d_runs, runs.data(), runs.size(), hipMemcpyHostToDevice);
    if (e != hipSuccess) rc = fail(DDD_ERR_HIP, "forcing runs upload: %s", hipGetErrorString(e));
  }
  if (rc) return rc;
  m->dp.runs = m->d_runs;
  m->dp.trig = m->d_trig;
  m->dp.frc = m->d_frc;
  m->dp.sp = m->d_sp;
  m->dp.forced = 1;
  m->dp.P = nparams;
  m->dp.n_k = n_k;
  m->dp.inv_P = 1.0f / (float)nparams;
  m->dp.inv_nk = 1.0f / (float)std::max(n_k, 1);
  m->dp.forcing_batch = batch;
  return DDD_OK;
}

int ddd_time_derivative(ddd_model* m, double t, const float* y, float* dydt, int batch,
                        void* stream) {
  int rc = check_batch(m, batch);
  if (rc) return rc;
  if (batch > 0 && (!y || !dydt)) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  ddd::SubstepArgs a{};
  a.t = t; a.y_in = y; a.c1 = 1.0f; a.y_out = dydt; a.batch = batch;
  return substep_entry(m, a, static_cast<hipStream_t>(stream));
}

int ddd_rk_substep(ddd_model* m, double t, const float* y_in, const float* y_base,
                   float c1, float* y_out, const float* acc_in, float c2,
                   float* acc_out, int batch, void* stream) {
  int rc = check_batch(m, batch);
  if (rc) return rc;
  if (batch > 0 && !y_in) return fail(DDD_ERR_INVALID_ARGUMENT, "y_in is NULL");
  if (batch > 0 && !y_out && !acc_out)
    return fail(DDD_ERR_INVALID_ARGUMENT, "both y_out and acc_out are NULL");
  ddd::SubstepArgs a{};
  a.t = t; a.y_in = y_in; a.y_base = y_base; a.c1 = c1; a.y_out = y_out;
  a.acc_in = acc_in; a.c2 = c2; a.acc_out = acc_out; a.batch = batch;
  return substep_entry(m, a, static_cast<hipStream_t>(stream));
}

int ddd_stream_fork(ddd_model* m, void* stream) {
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  if (m->spectral)
    return fail(DDD_ERR_UNSUPPORTED, "spectral models have no chained substeps");
  int rc = chain_close(m);   // a region that is still open ends here
  if (rc) return rc;
  m->chain.open = true;
  m->chain.stream = static_cast<hipStream_t>(stream);
  return DDD_OK;
}

int ddd_set_region_mode(ddd_model* m, int mode) {
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  if (mode != DDD_REGION_AUTO && mode != DDD_REGION_CHAINS && mode != DDD_REGION_RING)
    return fail(DDD_ERR_INVALID_ARGUMENT, "unknown region mode %d", mode);
  int rc = chain_close(m);
  if (rc) return rc;
  m->region_mode = mode;
  return DDD_OK;
}

int ddd_region_stats(const ddd_model* m, int64_t* ring_launches, int64_t* ring_commands) {
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  long long launches = 0, commands = 0;
  if (m->ring != nullptr) {
    std::lock_guard<std::mutex> lk(m->ring->mu);
    launches = (long long)m->ring->launches; commands = (long long)m->ring->commands;
  }
#ifdef DDD_RING_TRACE   // (variant build: where the stamps are)
  if (m->ring != nullptr) launches = (long long)(uintptr_t)(m->ring->status + 16);
#endif
  if (ring_launches != nullptr) *ring_launches = launches;
  if (ring_commands != nullptr) *ring_commands = commands;
  return DDD_OK;
}

int ddd_stream_join(ddd_model* m, void* stream) {
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  if (!m->chain.open) return DDD_OK;
  if (static_cast<hipStream_t>(stream) != m->chain.stream)
    return fail(DDD_ERR_INVALID_ARGUMENT,
                "ddd_stream_join: not the stream the region was opened on (ddd_stream_fork)");
  return chain_close(m);
}

int ddd_space_derivatives(ddd_model* m, const float* y, float* out, int batch,
                          void* stream) {
  if (m != nullptr) { int rcj = chain_close(m); if (rcj) return rcj; }
  int rc = check_batch(m, batch);
  if (rc) return rc;
  if (batch > 0 && (!y || !out)) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (!m->dp.fixed && m->dp.target != ddd::TARGET_COEFFICIENTS &&
      m->dp.target != ddd::TARGET_SPACE_DERIVATIVES)
    return fail(DDD_ERR_UNSUPPORTED, "model_target has no spatial derivatives");
  ddd::SubstepArgs a{};
  a.y_in = y; a.derivs_out = out; a.batch = batch;
  return launch_substep(m, a, static_cast<hipStream_t>(stream));
}

int ddd_coefficients(ddd_model* m, const float* y, float* out, int batch, void* stream) {
  if (m != nullptr) { int rcj = chain_close(m); if (rcj) return rcj; }
  int rc = check_batch(m, batch);
  if (rc) return rc;
  if (batch > 0 && (!y || !out)) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (!m->dp.fixed && m->dp.target != ddd::TARGET_COEFFICIENTS)
    return fail(DDD_ERR_UNSUPPORTED, "model_target does not produce coefficients");
  ddd::SubstepArgs a{};
  a.y_in = y; a.coeffs_out = out; a.batch = batch;
  return launch_substep(m, a, static_cast<hipStream_t>(stream));
}

int ddd_integrate_fixed(ddd_model* m, int scheme, int launch_mode, double t0, double dt,
                        int n_steps, int save_every, const float* y0, float* y_out,
                        int batch, void* stream_) {
  if (m != nullptr) { int rcj = chain_close(m); if (rcj) return rcj; }
  int rc = check_integrate_args(m, n_steps, save_every, y0, y_out, batch);
  if (rc) return rc;
  ddd::Tableau tab;
  rc = make_tableau(scheme, &tab);
  if (rc) return rc;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  if (launch_mode == DDD_LAUNCH_PERSISTENT) {
    ddd::IntegrateArgs a{};
    a.t0 = t0; a.dt = dt; a.n_steps = n_steps; a.save_every = save_every; a.tab = tab;
    a.sc = make_stage_consts(tab, dt);
    a.y0 = y0; a.y_out = y_out; a.batch = batch;
    return launch_integrate<float>(m, a, stream);
  }
  if (launch_mode != DDD_LAUNCH_PER_SUBSTEP && launch_mode != DDD_LAUNCH_PER_STEP)
    return fail(DDD_ERR_INVALID_ARGUMENT, "unknown launch_mode %d", launch_mode);
  if (batch == 0 || n_steps == 0) return DDD_OK;

  // One fused launch per substep; the state lives in HBM between launches.
  const size_t elems = (size_t)batch * m->dp.N;
  rc = ensure_scratch(m, 3 * elems);
  if (rc) return rc;
  float* ping = m->d_scratch;
  float* pong = m->d_scratch + elems;
  float* ystage = m->d_scratch + 2 * elems;
  const float h = (float)dt;
  const ddd::StageConsts stage_consts = make_stage_consts(tab, dt);

  m->last_launch_half = false; m->last_launch_quad = false; m->last_launch_split = false;
  // large ensembles: two half-ensembles side by side (plan_slabs)
  const SlabPlan plan = plan_slabs(m, batch);
  const int halves = plan.halves;
  const int* half_batch = plan.half_batch;
  const int* slab_first = plan.slab_first;
  int step_eq = -1;   // per-equation kernel for DDD_LAUNCH_PER_STEP
  if (m->kernel == DDD_KERNEL_MFMA && !m->explicit_kernel && !g_debug.no_spec) {
    const MfmaGeometry geo = mfma_geometry(m, batch);
    if (geo.wave_rows == 64 && launch_mode == DDD_LAUNCH_PER_STEP)
      step_eq = spec_equation(m, geo.rows);
  }
  if (launch_mode == DDD_LAUNCH_PER_STEP && ddd::stream::step_supports(m->dp) &&
      !m->explicit_kernel && !g_debug.no_stream && aligned16(y0) && aligned16(y_out) &&
      (elems % 4) == 0) {
    // fixed stencils, all stages of a step in one launch of the streaming kernel: the
    // stage inputs stay in the block's LDS tile, 8 B per grid point and step;
    // persistent blocks (eight per CU) walk over the tiles
    const long pts = (long)(ddd::stream::kStepTile / m->dp.N) * m->dp.N;
    const int tiles = (int)(((long)elems + pts - 1) / pts);
    const int grid = std::min(tiles, 2 * device_simds());
    const float* y = y0;
    size_t snap = 0;
    for (int step = 0; step < n_steps; ++step) {
      const bool saving = (step + 1) % save_every == 0;
      float* ynew = saving ? y_out + snap * elems : (y == ping ? pong : ping);
      ddd::StepArgs sa{};
      sa.t = t0 + (double)step * dt; sa.dt = dt; sa.tab = tab; sa.sc = stage_consts;
      sa.y_in = y; sa.y_out = ynew; sa.batch = batch;
      ddd::stream::launch_fixed_step(dim3(grid), stream, m->dp, sa, tiles);
      y = ynew;
      if (saving) ++snap;
    }
    DDD_HIP(hipGetLastError());
    m->last_batch = batch;
    m->last_launch_streamed = true; m->last_launch_lean = false;
    return DDD_OK;
  }
  hipStream_t lanes[kMaxParts] = {stream, stream, stream, stream};
  if (halves > 1) {
    rc = fork_lanes(m, stream, halves);
    if (rc) return rc;
    for (int i = 0; i < halves; ++i) lanes[i] = m->aux_stream[i];
  }
  size_t half_off[kMaxParts];
  for (int i = 0; i < kMaxParts; ++i) half_off[i] = (size_t)slab_first[i] * m->dp.N;
  const float* y = y0;
  size_t snap = 0;
#ifdef DDD_PROBES
  int walk_launch = 0;
#endif
  for (int step = 0; step < n_steps; ++step) {
    const double t = t0 + (double)step * dt;
    const bool saving = (step + 1) % save_every == 0;
    float* ynew = saving ? y_out + snap * elems : (y == ping ? pong : ping);
    if (step_eq >= 0) {
      // all stages in one launch per half-ensemble (step_multi_kernel)
      m->dp.dpp_rol = dpp_wave_rol_ok();
      const MfmaGeometry geo = mfma_geometry(m, batch);
      const int spg = geo.rows / m->dp.N;
      const int capacity = (geo.rows == 64 ? 2 * device_simds() : device_simds() / 2) / halves;
      for (int hf = 0; hf < halves; ++hf) {
        if (half_batch[hf] == 0) continue;
        ddd::DevParams dp = m->dp;
        if (slab_first[hf] != 0 && dp.forced) {
          dp.frc += (size_t)slab_first[hf] * dp.P;
          dp.runs += (size_t)slab_first[hf] * 8;
        }
        ddd::StepArgs sa{};
        sa.t = t; sa.dt = dt; sa.tab = tab; sa.sc = stage_consts;
        sa.y_in = y + half_off[hf]; sa.y_out = ynew + half_off[hf]; sa.batch = half_batch[hf];
        const int groups = (half_batch[hf] + spg - 1) / spg;
        const int grid = std::min(groups, capacity);
        const bool half = step_eq >= 0 && geo.rows == 64 && m->d_w_hidden_half != nullptr &&
                          m->d_w_final4_half != nullptr && m->d_w_t16 != nullptr && !g_debug.no_half;
        m->last_launch_half = half;
        if (half) { dp.w_hidden = m->d_w_hidden_half; dp.w_final4 = m->d_w_final4_half; dp.w_quad = m->d_w_t16; }
        switch (step_eq) {
#define DDD_STEP_CASE(EQ) \
          case EQ:                                                                              \
            if (half) ddd::launch::step_half_spec<EQ>(dp, sa, groups, grid, lanes[hf]);         \
            else ddd::launch::step_spec<EQ>(geo.rows, dp, sa, groups, grid, lanes[hf]);         \
            break;
          DDD_STEP_CASE(ddd::EQ_BURGERS) DDD_STEP_CASE(ddd::EQ_BURGERS_CONS)
          DDD_STEP_CASE(ddd::EQ_KDV) DDD_STEP_CASE(ddd::EQ_KDV_CONS)
          DDD_STEP_CASE(ddd::EQ_KS) DDD_STEP_CASE(ddd::EQ_KS_CONS)
#undef DDD_STEP_CASE
          default: break;
        }
      }
      if (hipGetLastError() != hipSuccess) {
        rc = fail(DDD_ERR_HIP, "step launch failed");
        break;
      }
      m->last_launch_streamed = false; m->last_launch_lean = false;
      y = ynew;
      if (saving) ++snap;
      continue;
    }
    const float* acc = nullptr;   // nullptr: accumulator still equals y
    for (int s = 0; s < tab.stages; ++s) {
      const bool last = s == tab.stages - 1;
      const bool accumulate = tab.b[s] != 0.0f || last;
      for (int hf = 0; hf < halves; ++hf) {
        const size_t off = half_off[hf];
        ddd::SubstepArgs a{};
        a.t = t + tab.c[s] * dt;
        a.y_in = (s == 0 ? y : ystage) + off;
        a.batch = half_batch[hf];
        if (!last) {   // next stage input  y + a_{s+1} h f
          a.y_base = y + off; a.c1 = tab.a[s + 1] * h; a.y_out = ystage + off;
        }
        if (accumulate) {
          a.acc_in = (acc != nullptr ? acc : y) + off;
          a.c2 = tab.b[s] * h;
          a.acc_out = ynew + off;
        }
#ifdef DDD_PROBES
        a.trace = reinterpret_cast<unsigned long long*>(g_debug.walk_trace_ptr);
        a.trace_row0 = walk_launch++ * ddd::kWalkTraceRows;
#endif
        rc = launch_substep(m, a, lanes[hf], slab_first[hf], halves);
        if (rc) break;
      }
      if (rc) break;
      if (accumulate) acc = ynew;
    }
    if (rc) break;   // (the join below still runs: the internal streams may be busy)
    y = ynew;
    if (saving) ++snap;
  }
  if (halves > 1) {
    const int rc_join = join_lanes(m, stream, halves);
    if (rc == DDD_OK) rc = rc_join;
    m->last_batch = batch;
  }
  return rc;
}

int ddd_integrate_fixed_f64(ddd_model* m, int scheme, double t0, double dt, int n_steps,
                            int save_every, const double* y0, double* y_out, int batch,
                            void* stream) {
  if (m != nullptr) { int rcj = chain_close(m); if (rcj) return rcj; }
  if (m != nullptr && m->spectral) {
    // float64 state AND right-hand side, one fused launch per substep
    if (batch < 0 || n_steps < 0 || save_every < 1)
      return fail(DDD_ERR_INVALID_ARGUMENT, "bad batch / n_steps / save_every");
    if (batch == 0 || n_steps == 0) return DDD_OK;
    if (y0 == nullptr || y_out == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
    ddd::Tableau tab;
    int rcs = make_tableau(scheme, &tab);
    if (rcs) return rcs;
    double b64[ddd::kMaxStages];
    tableau_weights_f64(scheme, b64);
    const size_t elems = (size_t)batch * m->sp.N;
    if (m->scratch64_doubles < 3 * elems) {
      if (m->d_scratch64 != nullptr) (void)hipFree(m->d_scratch64);
      m->d_scratch64 = nullptr; m->scratch64_doubles = 0;
      DDD_HIP(hipMalloc(reinterpret_cast<void**>(&m->d_scratch64), 3 * elems * sizeof(double)));
      m->scratch64_doubles = 3 * elems;
    }
    double* ping = m->d_scratch64;
    double* pong = ping + elems;
    double* ystage = ping + 2 * elems;
    const double* y = y0;
    size_t snap = 0;
    for (int step = 0; step < n_steps; ++step) {
      const bool saving = (step + 1) % save_every == 0;
      double* ynew = saving ? y_out + snap * elems : (y == ping ? pong : ping);
      const double* acc = nullptr;
      for (int s = 0; s < tab.stages; ++s) {
        ddd::spectral::SubstepArgs64 a{};
        a.y_in = s == 0 ? y : ystage;
        a.batch = batch;
        const bool last = s == tab.stages - 1;
        if (!last) { a.y_base = y; a.c1 = (double)tab.a[s + 1] * dt; a.y_out = ystage; }
        if (tab.b[s] != 0.0f || last) {
          a.acc_in = acc != nullptr ? acc : y;
          a.c2 = b64[s] * dt;
          a.acc_out = ynew;
          acc = ynew;
        }
        rcs = launch_spectral(m, a, static_cast<hipStream_t>(stream));
        if (rcs) return rcs;
      }
      y = ynew;
      if (saving) ++snap;
    }
    return DDD_OK;
  }
  int rc = check_integrate_args(m, n_steps, save_every, y0, y_out, batch);
  if (rc) return rc;
  ddd::IntegrateArgs a{};
  rc = make_tableau(scheme, &a.tab);
  if (rc) return rc;
  a.t0 = t0; a.dt = dt; a.n_steps = n_steps; a.save_every = save_every;
  a.sc = make_stage_consts(a.tab, dt);
  a.y0 = y0; a.y_out = y_out; a.batch = batch;
  return launch_integrate<double>(m, a, static_cast<hipStream_t>(stream));
}

int ddd_integrate_adaptive_f64(ddd_model* m, const double* times, int n_times, double rtol,
                               double atol, double max_step, long long max_attempts,
                               const double* y0, double* y_out, int32_t* nfev,
                               int32_t* status, int batch, void* stream_) {
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  int rc = chain_close(m);
  if (rc) return rc;
  rc = check_batch(m, batch, m->spectral);
  if (rc) return rc;
  if (times == nullptr || n_times < 1)
    return fail(DDD_ERR_INVALID_ARGUMENT, "times must hold at least one value");
  if (n_times >= (1 << 28))   // (the packed per-sample controllers keep the output index in 28 bits)
    return fail(DDD_ERR_INVALID_ARGUMENT, "n_times = %d: at most 2^28 - 1 output times", n_times);
  for (int i = 0; i < n_times; ++i)
    if (!std::isfinite(times[i]) || (i > 0 && !(times[i] > times[i - 1])))
      return fail(DDD_ERR_INVALID_ARGUMENT, "times must be finite and strictly increasing");
  if (!(rtol > 0.0) || !(atol >= 0.0) || !(max_step > 0.0))
    return fail(DDD_ERR_INVALID_ARGUMENT, "rtol and max_step must be positive, atol >= 0");
  if (batch > 0 && (!y0 || !y_out || !nfev || !status))
    return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (m->spectral && is_forced_family(m->cfg.equation))
    return fail(DDD_ERR_UNSUPPORTED,
                "the spectral adaptive kernel carries no forcing term: Burgers' forcing(t) "
                "(equations.py:276-277) would be dropped; integrate it with the host driver "
                "(integrate.odeint over ddd_time_derivative_f64 + finalize_time_derivative)");
  if (batch == 0) return DDD_OK;
  hipStream_t stream = static_cast<hipStream_t>(stream_);
  // the caller's array may be gone before the copy runs: upload from a page-locked
  // copy the model owns until the launch that reads it has finished
  ddd_model::TimeSlot& slot = m->time_slot[m->next_time_slot];
  m->next_time_slot = (m->next_time_slot + 1) % ddd_model::kTimeSlots;
  if (slot.in_flight) {   // only when kTimeSlots adaptive calls are still queued
    DDD_HIP(hipEventSynchronize(slot.done));
    slot.in_flight = false;
  }
  if (slot.capacity < (size_t)n_times) {
    if (slot.host != nullptr) (void)hipHostFree(slot.host);
    free_dev(slot.dev);
    slot.host = nullptr; slot.dev = nullptr; slot.capacity = 0;
    DDD_HIP(hipHostMalloc(reinterpret_cast<void**>(&slot.host), (size_t)n_times * sizeof(double),
                          hipHostMallocDefault));
    DDD_HIP(hipMalloc(reinterpret_cast<void**>(&slot.dev), (size_t)n_times * sizeof(double)));
    slot.capacity = (size_t)n_times;
  }
  if (slot.done == nullptr)
    DDD_HIP(hipEventCreateWithFlags(&slot.done, hipEventDisableTiming));
  std::memcpy(slot.host, times, (size_t)n_times * sizeof(double));
  DDD_HIP(hipMemcpyAsync(slot.dev, slot.host, (size_t)n_times * sizeof(double),
                         hipMemcpyHostToDevice, stream));
  // the slot is busy from HERE on (the copy reads the pinned buffer), whatever happens to
  // the launch below: an error return between the copy and the launch must not leave a
  // slot that looks free while the copy is still pending
  DDD_HIP(hipEventRecord(slot.done, stream));
  slot.in_flight = true;
  ddd::AdaptiveArgs a{};
  // ... and is free again once the launch enqueued below has run
  const auto enqueued = [&]() -> int {
    DDD_HIP(hipGetLastError());
    DDD_HIP(hipEventRecord(slot.done, stream));
    slot.in_flight = true;
    return DDD_OK;
  };
  a.times = slot.dev; a.n_times = n_times;
  a.rtol = rtol; a.atol = atol; a.max_step = max_step;
  a.y0 = y0; a.y_out = y_out; a.nfev = nfev; a.status = status; a.batch = batch;
  if (max_attempts <= 0) {
    // generous default: a thousand times the attempts of a run at max_step
    const double span = times[n_times - 1] - times[0];
    max_attempts = (long long)std::min(1e15, std::ceil(span / max_step) * 1000.0 + 100000.0);
  }
  a.max_attempts = max_attempts;
  m->last_batch = batch;
  m->last_launch_streamed = false; m->last_launch_lean = false;
  m->last_launch_split = false; m->last_launch_quad = false; m->last_launch_half = false;
  if (m->spectral) {
    // float64 right-hand side (SpectralDifferentiator), one workgroup per sample
    const size_t lds = ddd::spectral::lds_bytes(m->sp);
    const int pts = (m->sp.N + ddd::spectral::kThreads - 1) / ddd::spectral::kThreads;
#define DDD_SPECTRAL_ADAPTIVE(PTS)                                                           \
    do {                                                                                     \
      DDD_HIP(hipFuncSetAttribute(                                                           \
          reinterpret_cast<const void*>(ddd::spectral::adaptive_kernel<PTS>),                \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                            \
      hipLaunchKernelGGL((ddd::spectral::adaptive_kernel<PTS>), dim3(batch),                 \
                         dim3(ddd::spectral::kThreads), lds, stream, m->sp, a);              \
    } while (0)
    if (pts <= 1) DDD_SPECTRAL_ADAPTIVE(1);
    else if (pts <= 2) DDD_SPECTRAL_ADAPTIVE(2);
    else if (pts <= 4) DDD_SPECTRAL_ADAPTIVE(4);
    else DDD_SPECTRAL_ADAPTIVE(8);
#undef DDD_SPECTRAL_ADAPTIVE
    return enqueued();
  }
  if (use_weno_kernel(m)) {
    // the WENO5 + Godunov-flux exact solver: one wavefront and one controller per sample
    ddd::launch::weno_adaptive(m->dp, a, stream);
    return enqueued();
  }
  if (m->kernel != DDD_KERNEL_MFMA) {
    // generic right-hand side (nets the MFMA path does not carry, WENO on grids that are
    // not 64 x {1, 2, 4, 8} points): one workgroup and one controller per sample
    const size_t lds = ddd::generic::adaptive_lds_bytes(m->dp);
    if (lds > 160 * 1024)
      return fail(DDD_ERR_UNSUPPORTED,
                  "generic adaptive kernel needs %zu B of LDS per sample (limit 163840)", lds);
    DDD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ddd::generic::adaptive_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ddd::generic::adaptive_kernel, dim3(batch), dim3(ddd::generic::kThreads),
                       lds, stream, m->dp, a);
    return enqueued();
  }
  m->dp.dpp_rol = dpp_wave_rol_ok();
  MfmaGeometry geo = mfma_geometry(m, batch);
  const bool want_quad = geo.rows == 64 && (geo.wave_rows == 16 || (geo.wave_rows == 64 && m->force_rows == 0));
  if (geo.wave_rows != 64) geo = {64, 64};   // the two-wave split has no adaptive instantiation
  const int spg = geo.rows / m->dp.N;
  const int blocks = (batch + spg - 1) / spg;
  // small ensembles: every 64-row group on four 16-row wavefronts (rhs_mfma.h kQuad), as
  // launch_integrate chooses it for the fixed-step integrators -- the reference's callers
  // integrate tens to hundreds of samples (scripts/run_evaluation.py:212-221)
  m->last_launch_quad = false; m->last_launch_half = false;
  if (want_quad && m->dp.w_quad != nullptr && spec_equation(m, 64) >= 0 &&
      (m->force_rows == 16 || (m->split_auto && !m->explicit_kernel && 2 * blocks <= device_simds()))) {
    m->last_launch_quad = true;
    switch (spec_equation(m, 64)) {
#define DDD_ADAPTIVE_QUAD(EQ) \
      case EQ: ddd::launch::adaptive_quad_spec<EQ>(m->dp, a, blocks, stream); break;
      DDD_ADAPTIVE_QUAD(ddd::EQ_BURGERS)
      DDD_ADAPTIVE_QUAD(ddd::EQ_BURGERS_CONS)
      DDD_ADAPTIVE_QUAD(ddd::EQ_KDV)
      DDD_ADAPTIVE_QUAD(ddd::EQ_KDV_CONS)
      DDD_ADAPTIVE_QUAD(ddd::EQ_KS)
      DDD_ADAPTIVE_QUAD(ddd::EQ_KS_CONS)
#undef DDD_ADAPTIVE_QUAD
      default: break;
    }
    return enqueued();
  }
  // nets of up to 16 filters: the block-diagonal tower (one-wave groups)
  const bool half = geo.rows == 64 && m->d_w_hidden_half != nullptr && m->d_w_final4_half != nullptr &&
                    spec_equation(m, 64) >= 0 && !g_debug.no_half;
  m->last_launch_half = half;
  ddd::DevParams dp_half = m->dp;
  if (half) { dp_half.w_hidden = m->d_w_hidden_half; dp_half.w_final4 = m->d_w_final4_half; dp_half.w_quad = m->d_w_t16; }
  switch (spec_equation(m, geo.rows)) {
#define DDD_ADAPTIVE_CASE(EQ) \
    case EQ:                                                                        \
      if (half) ddd::launch::adaptive_half_spec<EQ>(dp_half, a, blocks, stream);    \
      else ddd::launch::adaptive_spec<EQ>(geo.rows, m->dp, a, blocks, stream);      \
      break;
    DDD_ADAPTIVE_CASE(ddd::EQ_BURGERS)
    DDD_ADAPTIVE_CASE(ddd::EQ_BURGERS_CONS)
    DDD_ADAPTIVE_CASE(ddd::EQ_KDV)
    DDD_ADAPTIVE_CASE(ddd::EQ_KDV_CONS)
    DDD_ADAPTIVE_CASE(ddd::EQ_KS)
    DDD_ADAPTIVE_CASE(ddd::EQ_KS_CONS)
#undef DDD_ADAPTIVE_CASE
    default:
      if (m->big()) {
#define DDD_BIG_ADAPTIVE(K, CB)                                                           \
        if (m->tower_k == K && m->tower_cb == CB) {                                       \
          if (geo.rows == 64) ddd::launch::adaptive_big_unit<K, CB, 64>(m->dp, a, blocks, stream); \
          else ddd::launch::adaptive_big_unit<K, CB, 256>(m->dp, a, blocks, stream);      \
        }
        DDD_FOR_EACH_BIG_TOWER(DDD_BIG_ADAPTIVE)
#undef DDD_BIG_ADAPTIVE
      } else if (m->wide) {
        if (geo.rows == 64) ddd::launch::adaptive_wide_unit<64>(m->dp, a, blocks, stream);
        else ddd::launch::adaptive_wide_unit<256>(m->dp, a, blocks, stream);
      } else if (geo.rows == 64) {
        ddd::launch::adaptive_runtime_unit<64>(m->dp, a, blocks, stream);
      } else {
        ddd::launch::adaptive_runtime_unit<256>(m->dp, a, blocks, stream);
      }
  }
  return enqueued();
}

int ddd_circulant_apply_f64(const double* kernel, const double* in, double* out, int rows,
                            int n, void* stream) {
  if (!kernel || !in || !out) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (rows < 0 || n < 1 || n > ddd::spectral::kMaxPoints)
    return fail(DDD_ERR_INVALID_ARGUMENT, "rows >= 0 and 1 <= n <= %d required",
                ddd::spectral::kMaxPoints);
  if (rows == 0) return DDD_OK;
  const size_t lds = 2 * (size_t)n * sizeof(double);
  hipLaunchKernelGGL(ddd::spectral::circulant_apply_kernel, dim3(rows),
                     dim3(ddd::spectral::kThreads), lds, static_cast<hipStream_t>(stream),
                     kernel, in, out, n);
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int ddd_conv1d_periodic(const float* in, const float* filters, const float* bias,
                        float* out, int batch, int n, int cin, int cout, int k,
                        int center, int activation, void* stream) {
  if (!in || !filters || !out) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (batch < 0 || n < 1 || cin < 1 || cout < 1 || k < 1)
    return fail(DDD_ERR_INVALID_ARGUMENT, "bad conv1d shape");
  if (activation < -1 || activation > DDD_ACT_ELU)
    return fail(DDD_ERR_INVALID_ARGUMENT, "unknown activation %d", activation);
  const long total = (long)batch * n * cout;
  if (total == 0) return DDD_OK;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(ddd::ops::conv1d_periodic_kernel, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, filters, bias, out, batch, n,
                     cin, cout, k, center ? k / 2 : 0, activation);
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int ddd_pad_periodic(const float* in, float* out, int batch, int n, int c, int padding,
                     int center, void* stream) {
  if (!in || !out) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (batch < 0 || n < 1 || c < 1 || padding < 0)
    return fail(DDD_ERR_INVALID_ARGUMENT, "bad pad_periodic shape");
  const long total = (long)batch * (n + padding) * c;
  if (total == 0) return DDD_OK;
  const int left = center ? (padding + 1) / 2 : 0;   // layers.py:76-79
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(ddd::ops::pad_periodic_kernel, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, batch, n, c, padding, left);
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int ddd_extract_patches(const float* in, float* out, int batch, int n, int size, void* stream) {
  if (!in || !out) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (batch < 0 || n < 1 || size < 1) return fail(DDD_ERR_INVALID_ARGUMENT, "bad sizes");
  if (batch == 0) return DDD_OK;
  const long total = (long)batch * n * size;
  const int blocks = (int)std::min<long>((total + 255) / 256, 65535);
  hipLaunchKernelGGL(ddd::ops::extract_patches_kernel, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), in, out, batch, n, size);
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int ddd_apply_coefficients(const float* coefficients, const float* in, float* out, int batch,
                           int n, int d, int g, void* stream) {
  if (!coefficients || !in || !out) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (batch < 0 || n < 1 || d < 1 || g < 1) return fail(DDD_ERR_INVALID_ARGUMENT, "bad sizes");
  if (batch == 0) return DDD_OK;
  const long total = (long)batch * n * d;
  const int blocks = (int)std::min<long>((total + 255) / 256, 65535);
  hipLaunchKernelGGL(ddd::ops::apply_coefficients_kernel, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), coefficients, in, out, batch, n, d, g);
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int ddd_apply_space_derivatives(int equation, const float* derivatives, const float* y,
                                float* out, int batch, int n, int d, double eta, double dx,
                                void* stream) {
  if (!derivatives || !y || !out) return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (equation < DDD_EQ_BURGERS || equation > DDD_EQ_KS_GODUNOV)
    return fail(DDD_ERR_INVALID_ARGUMENT, "unknown equation %d", equation);
  if (batch < 0 || n < 2 || d < 1 || d > DDD_MAX_DERIVATIVES || !(dx > 0))
    return fail(DDD_ERR_INVALID_ARGUMENT, "bad sizes");
  if (batch == 0) return DDD_OK;
  const bool flux_form = equation == DDD_EQ_BURGERS_CONSERVATIVE ||
                         equation == DDD_EQ_KDV_CONSERVATIVE ||
                         equation == DDD_EQ_KS_CONSERVATIVE || equation >= DDD_EQ_BURGERS_GODUNOV;
  const size_t lds = (size_t)n * sizeof(float);
  if (lds > 64 * 1024) return fail(DDD_ERR_UNSUPPORTED, "num_points too large");
  hipLaunchKernelGGL(ddd::ops::apply_space_derivatives_kernel, dim3(batch), dim3(256), lds,
                     static_cast<hipStream_t>(stream), derivatives, y, out, equation, n, d,
                     (float)eta, (float)(1.0 / dx), flux_form ? 1 : 0);
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int ddd_polynomial_accuracy_apply(const float* inputs, const float* nullspace,
                                  const float* bias, float* out, int64_t m,
                                  int input_size, int g, void* stream) {
  if (!inputs || !nullspace || !bias || !out)
    return fail(DDD_ERR_INVALID_ARGUMENT, "NULL array");
  if (m < 0 || input_size < 1 || g < 1)
    return fail(DDD_ERR_INVALID_ARGUMENT, "bad polynomial_accuracy shape");
  if (m == 0) return DDD_OK;
  const long total = (long)m * g;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  hipLaunchKernelGGL(ddd::ops::polynomial_accuracy_kernel, dim3(blocks), dim3(256), 0,
                     static_cast<hipStream_t>(stream), inputs, nullspace, bias, out,
                     (long)m, input_size, g);
  DDD_HIP(hipGetLastError());
  return DDD_OK;
}

int ddd_set_kernel(ddd_model* m, int kind) {
  if (m != nullptr) { int rcj = chain_close(m); if (rcj) return rcj; }
  if (m == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "model is NULL");
  if (m->spectral) return fail(DDD_ERR_UNSUPPORTED, "spectral models have one kernel");
  switch (kind) {
    case DDD_KERNEL_AUTO:
      m->kernel = m->mfma_ok ? DDD_KERNEL_MFMA : DDD_KERNEL_GENERIC;
      m->force_rows = 0;
      m->explicit_kernel = false;
      return DDD_OK;
    case DDD_KERNEL_GENERIC:
      m->kernel = DDD_KERNEL_GENERIC;
      m->explicit_kernel = true;
      return DDD_OK;
    case DDD_KERNEL_MFMA:
    case DDD_KERNEL_MFMA_ROWS64:
    case DDD_KERNEL_MFMA_ROWS256:
    case DDD_KERNEL_MFMA_ROWS64_W32:
    case DDD_KERNEL_MFMA_ROWS64_W16:
      if (!m->mfma_ok)
        return fail(DDD_ERR_UNSUPPORTED, "MFMA path unavailable for this model: %s",
                    m->mfma_reason.c_str());
      if (kind == DDD_KERNEL_MFMA_ROWS64_W32 && (m->wide || m->big()))
        return fail(DDD_ERR_UNSUPPORTED,
                    "the two-wave split exists for the 5-tap x 32-filter tower with stencils <= 8 "
                    "points / <= 16 channels only");
      if (kind == DDD_KERNEL_MFMA_ROWS64_W16 && m->dp.w_quad == nullptr)
        return fail(DDD_ERR_UNSUPPORTED,
                    "four 16-row wavefronts per group exist for three-layer 5-tap x 32-filter nets "
                    "with <= 16 output channels only");
      if ((kind == DDD_KERNEL_MFMA_ROWS64 || kind == DDD_KERNEL_MFMA_ROWS64_W32 ||
           kind == DDD_KERNEL_MFMA_ROWS64_W16) &&
          !(m->dp.N <= 64 && 64 % m->dp.N == 0))
        return fail(DDD_ERR_UNSUPPORTED,
                    "64-row workgroups need num_points to divide 64 (got %d)", m->dp.N);
      m->kernel = DDD_KERNEL_MFMA;
      m->explicit_kernel = true;
      m->force_rows = kind == DDD_KERNEL_MFMA_ROWS64 ? 64
                      : kind == DDD_KERNEL_MFMA_ROWS64_W32 ? 32
                      : kind == DDD_KERNEL_MFMA_ROWS64_W16 ? 16
                      : kind == DDD_KERNEL_MFMA_ROWS256 ? 256 : 0;
      return DDD_OK;
    default:
      return fail(DDD_ERR_INVALID_ARGUMENT, "unknown kernel kind %d", kind);
  }
}

const char* ddd_kernel_name(const ddd_model* m) {
  if (m == nullptr) return "";
  if (m->spectral) return "spectral_f64";
  if (m->last_launch_streamed) return "stream_fixed";
  if (m->last_launch_lean) return "valu_f32_lean";
  if (use_weno_kernel(m)) return "valu_f32_weno";
  if (m->kernel != DDD_KERNEL_MFMA) return "generic";
  const MfmaGeometry geo = mfma_geometry(m, m->last_batch > 0 ? m->last_batch : 1 << 30);
  if (geo.rows == 256) return "mfma_f32_r256";
  if (m->last_launch_quad) return "mfma_f32_r64w16";
  if (m->last_launch_half) return "mfma_f32_r64h16";
  return (geo.wave_rows == 32 || m->last_launch_split) ? "mfma_f32_r64w32" : "mfma_f32_r64";
}

int64_t ddd_fma_per_point(const ddd_model* m) { return m ? m->fma_per_point : 0; }

#ifdef DDD_PROBES
// Profiling / A-B switches (not part of the product API; every change is
// logged).  Names: no_fold (takes effect at ddd_model_create), no_spec,
// no_stream, prio_split, stagger, substep_parts, ablate, trace_ptr, walk_trace_ptr; "reset" puts
// every switch back to its default (the switches are process-global).
DDD_API int ddd_debug_set_option(const char* name, long long value) {
  if (name == nullptr) return fail(DDD_ERR_INVALID_ARGUMENT, "name is NULL");
  const std::string key(name);
  if (key == "no_fold") g_debug.no_fold = (int)value;
  else if (key == "no_spec") g_debug.no_spec = (int)value;
  else if (key == "no_stream") g_debug.no_stream = (int)value;
  else if (key == "no_lean") g_debug.no_lean = (int)value;
  else if (key == "no_weno") g_debug.no_weno = (int)value;
  else if (key == "no_fft") g_debug.no_fft = (int)value;
  else if (key == "no_half") g_debug.no_half = (int)value;
  else if (key == "prio_split") g_debug.prio_split = (int)value;
  else if (key == "stagger") g_debug.stagger = (int)value;
  else if (key == "substep_parts") g_debug.substep_parts = (int)value;
  else if (key == "reset") g_debug = DebugOptions{};   // every switch back to its default
  else if (key == "ablate") g_debug.ablate = (int)value;
  else if (key == "trace_ptr") g_debug.trace_ptr = (unsigned long long)value;
  else if (key == "walk_trace_ptr") g_debug.walk_trace_ptr = (unsigned long long)value;
  else return fail(DDD_ERR_INVALID_ARGUMENT, "unknown debug option '%s'", name);
  fprintf(stderr, "libddd1d: debug option %s = %lld%s\n", name, value,
          key == "ablate" && value != 0 ? " (kernel phases skipped: results are WRONG)" : "");
  return DDD_OK;
}

DDD_API int ddd_debug_hwid(unsigned* out_host, int blocks, int spin) {
  unsigned* d = nullptr;
  DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d), (size_t)blocks * 4 * sizeof(unsigned)));
  hipLaunchKernelGGL(ddd::ops::hwid_probe_kernel, dim3(blocks), dim3(64), 0, nullptr, d, spin);
  DDD_HIP(hipGetLastError());
  DDD_HIP(hipMemcpy(out_host, d, (size_t)blocks * 4 * sizeof(unsigned), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return DDD_OK;
}

// Issue-port sharing probe (ops.h: issue_share_probe_kernel).  Returns mean
// s_memtime ticks per operation for the MFMA streamers (slot 0) and the
// workers (slot 1) over `blocks` workgroups.
DDD_API int ddd_debug_issue_share(int mfma_kind, int work_kind, int blocks, int iters, int prio,
                          double* mfma_ticks_per_op, double* work_ticks_per_op) {
  unsigned long long* d = nullptr;
  const size_t words = (size_t)blocks * 8 * 4;
  DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d), words * sizeof(unsigned long long)));
  DDD_HIP(hipMemset(d, 0, words * sizeof(unsigned long long)));
#define DDD_SHARE(M, W)                                                                  \
  if (mfma_kind == M && work_kind == W)                                                  \
    hipLaunchKernelGGL((ddd::ops::issue_share_probe_kernel<M, W>), dim3(blocks), dim3(512), \
                       0, nullptr, d, iters, 1.0f, prio);
  for (int rep = 0; rep < 2; ++rep) {
    DDD_SHARE(0, 1) DDD_SHARE(0, 2) DDD_SHARE(0, 3) DDD_SHARE(0, 4)
    DDD_SHARE(1, 0) DDD_SHARE(1, 1) DDD_SHARE(1, 2) DDD_SHARE(1, 3) DDD_SHARE(1, 4)
    DDD_SHARE(2, 0) DDD_SHARE(2, 1) DDD_SHARE(2, 2) DDD_SHARE(2, 3) DDD_SHARE(2, 4)
    DDD_SHARE(3, 0) DDD_SHARE(3, 1) DDD_SHARE(3, 2) DDD_SHARE(4, 0) DDD_SHARE(4, 1) DDD_SHARE(4, 2)
    DDD_SHARE(5, 0) DDD_SHARE(5, 1) DDD_SHARE(5, 2) DDD_SHARE(5, 3) DDD_SHARE(6, 0) DDD_SHARE(6, 1) DDD_SHARE(6, 2)
    DDD_SHARE(7, 0) DDD_SHARE(7, 1) DDD_SHARE(7, 2) DDD_SHARE(8, 0) DDD_SHARE(8, 1) DDD_SHARE(8, 2) DDD_SHARE(8, 3)
    DDD_HIP(hipDeviceSynchronize());
  }
#undef DDD_SHARE
  DDD_HIP(hipGetLastError());
  std::vector<unsigned long long> h(words);
  DDD_HIP(hipMemcpy(h.data(), d, words * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  double sm = 0, sw = 0;
  long nm = 0, nw = 0;
  for (size_t g = 0; g < (size_t)blocks * 8; ++g) {
    const unsigned slot = (unsigned)(h[g * 4 + 1] & 0xff);
    if (slot == 0) { sm += (double)h[g * 4]; ++nm; } else { sw += (double)h[g * 4]; ++nw; }
  }
  const double mops = (double)iters * ((mfma_kind == 2 || mfma_kind >= 7) ? 256.0 : 64.0);
  const double wops = (double)iters * 64.0;
  *mfma_ticks_per_op = nm ? sm / nm / mops : 0.0;
  *work_ticks_per_op = nw ? sw / nw / wops : 0.0;
  return DDD_OK;
}

DDD_API int ddd_debug_mfma_rate(int chains, int is32, int blocks, int iters, double* ticks_per_mfma,
                        double* wall_ns_per_mfma) {
  unsigned long long* d = nullptr;
  DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d), (size_t)blocks * 2 * sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  DDD_HIP(hipEventCreate(&e0));
  DDD_HIP(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    DDD_HIP(hipEventRecord(e0, nullptr));
#define DDD_RATE(C, W) hipLaunchKernelGGL((ddd::ops::mfma_rate_probe_kernel<C, W>), dim3(blocks), dim3(64), 0, nullptr, d, iters, 1.0f)
#define DDD_RATE4(C) hipLaunchKernelGGL((ddd::ops::mfma4_rate_probe_kernel<C>), dim3(blocks), dim3(64), 0, nullptr, d, iters, 1.0f)
    if (is32 == 2) {   // v_mfma_f32_4x4x1_16b_f32, 24 per iteration
      if (chains == 1) DDD_RATE4(1); else if (chains == 2) DDD_RATE4(2);
      else if (chains == 3) DDD_RATE4(3); else DDD_RATE4(4);
    } else if (is32) { if (chains == 1) DDD_RATE(1, true); else if (chains == 2) DDD_RATE(2, true); else DDD_RATE(4, true); }
    else { if (chains == 1) DDD_RATE(1, false); else if (chains == 2) DDD_RATE(2, false); else DDD_RATE(4, false); }
#undef DDD_RATE4
#undef DDD_RATE
    DDD_HIP(hipEventRecord(e1, nullptr));
    DDD_HIP(hipEventSynchronize(e1));
  }
  float ms = 0.0f;
  DDD_HIP(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h((size_t)blocks * 2);
  DDD_HIP(hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  double sum = 0.0;
  for (int b = 0; b < blocks; ++b) sum += (double)h[(size_t)b * 2];
  const double per_iter = is32 == 2 ? 24.0 : 8.0;
  *ticks_per_mfma = sum / blocks / ((double)iters * per_iter);
  *wall_ns_per_mfma = (double)ms * 1e6 / ((double)iters * per_iter);
  return DDD_OK;
}
#endif  // DDD_PROBES

int ddd_selftest_mfma_layout(void) {
  float* d32 = nullptr;
  float* d16 = nullptr;
  DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d32), 32 * 32 * sizeof(float)));
  DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d16), 16 * 16 * sizeof(float)));
  hipLaunchKernelGGL(ddd::ops::mfma_layout_probe_kernel, dim3(1), dim3(64), 0, nullptr,
                     d32, d16);
  DDD_HIP(hipGetLastError());
  std::vector<float> h32(32 * 32), h16(16 * 16);
  DDD_HIP(hipMemcpy(h32.data(), d32, h32.size() * sizeof(float), hipMemcpyDeviceToHost));
  DDD_HIP(hipMemcpy(h16.data(), d16, h16.size() * sizeof(float), hipMemcpyDeviceToHost));
  (void)hipFree(d32);
  (void)hipFree(d16);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      const float want = (float)(i + 1) * (float)(64 * (j + 1)) +
                         (float)(1000 + i) * (float)(3 * j + 7);
      if (h32[i * 32 + j] != want)
        return fail(DDD_ERR_UNSUPPORTED,
                    "mfma_f32_32x32x2 layout mismatch at D[%d][%d]: got %g want %g", i, j,
                    h32[i * 32 + j], want);
    }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      float want = 0.0f;
      for (int k = 0; k < 4; ++k)
        want = fmaf((float)((k + 1) * 100 + i), (float)((k + 2) * (j + 1)), want);
      if (h16[i * 16 + j] != want)
        return fail(DDD_ERR_UNSUPPORTED,
                    "mfma_f32_16x16x4 layout mismatch at D[%d][%d]: got %g want %g", i, j,
                    h16[i * 16 + j], want);
    }
  {
    // 4x4x1 (16 blocks) with the A block broadcast: register r of lane l must be
    // A(lane 4 abid + r) * B(lane l) + C   (ops.h: mfma4_bcast)
    float* d4 = nullptr;
    DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d4), 16 * 4 * 64 * sizeof(float)));
    hipLaunchKernelGGL(ddd::ops::mfma4_layout_probe_kernel, dim3(1), dim3(64), 0, nullptr, d4);
    DDD_HIP(hipGetLastError());
    std::vector<float> h4(16 * 4 * 64);
    DDD_HIP(hipMemcpy(h4.data(), d4, h4.size() * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d4);
    for (int abid = 0; abid < 16; ++abid)
      for (int r = 0; r < 4; ++r)
        for (int l = 0; l < 64; ++l) {
          const float want = fmaf((float)(1000 * abid + r + 1), (float)(l + 1), 0.5f);
          const float got = h4[(abid * 4 + r) * 64 + l];
          if (got != want)
            return fail(DDD_ERR_UNSUPPORTED,
                        "mfma_f32_4x4x1 broadcast layout mismatch at abid %d reg %d lane %d: "
                        "got %g want %g", abid, r, l, got, want);
        }
  }
  {
    // relu as the [0, 1] output clamp of a packed add on activations scaled by
    // 2^-kReluShift (rhs_mfma.h: activate16), and the constant-lane-mask select of the
    // input layer (upper_half_one)
    std::vector<float> h_in(128);
    const float dn = std::ldexp(1.0f, -ddd::kReluShift);
    for (int i = 0; i < 128; ++i) {
      const float mag = std::ldexp(1.0f + 0.0078125f * (float)i, (i * 7) % 120 - 60);   // 2^-60 .. 2^59
      h_in[i] = ((i % 3) == 1 ? -mag : mag) * dn;
    }
    h_in[5] = 0.0f; h_in[6] = -0.0f; h_in[7] = std::nanf(""); h_in[8] = 2.0f;   // (> 1: saturates)
    h_in[9] = -INFINITY; h_in[10] = INFINITY; h_in[11] = 1.0f;
    float *d_in = nullptr, *d_out = nullptr;
    DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d_in), 128 * sizeof(float)));
    DDD_HIP(hipMalloc(reinterpret_cast<void**>(&d_out), 192 * sizeof(float)));
    DDD_HIP(hipMemcpy(d_in, h_in.data(), 128 * sizeof(float), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(ddd::ops::relu_clamp_probe_kernel, dim3(1), dim3(64), 0, nullptr, d_in, d_out);
    DDD_HIP(hipGetLastError());
    std::vector<float> h_out(192);
    DDD_HIP(hipMemcpy(h_out.data(), d_out, 192 * sizeof(float), hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 2; ++e) {
        const float x = h_in[2 * l + e], got = h_out[64 * e + l];
        const float want = std::isnan(x) ? 0.0f : x < 0.0f ? 0.0f : x > 1.0f ? 1.0f : x;
        if (!(got == want) || std::signbit(got))
          return fail(DDD_ERR_UNSUPPORTED,
                      "v_pk_add_f32 ... clamp is not the [0, 1] clamp the relu assumes: "
                      "input %g (element %d of pair %d) gave %g, want %g", x, e, l, got, want);
      }
    for (int l = 0; l < 64; ++l)
      if (h_out[128 + l] != (l < 32 ? (float)l : 1.0f))
        return fail(DDD_ERR_UNSUPPORTED, "constant-lane-mask select: lane %d gave %g", l,
                    h_out[128 + l]);
  }
  // informational: the kernels fall back to ds_bpermute when this is 0
  (void)dpp_wave_rol_ok();
  return DDD_OK;
}

}  // extern "C"
