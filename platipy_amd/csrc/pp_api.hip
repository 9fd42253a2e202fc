// platipy_amd/csrc/pp_api.hip -- context management and host-side helpers of the C ABI
// (include/platipy_amd.h).
#include <algorithm>
#include <atomic>
#include <chrono>

#include "pp_internal.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

// ---------------------------------------------------------------------------------------
// optional HIP-event profiler

struct pp_prof_span {
  int slot;
  hipEvent_t a, b;
};
struct pp_profiler {
  std::vector<std::string> names;
  std::vector<pp_prof_span> spans;   // recorded since the last read
  std::vector<hipEvent_t> pool;      // idle events
  int open_slot = -1;
  hipEvent_t open_a = nullptr;
  // Sampling: a pair of events costs the stream ~7 us (3.7 us each: measured as profiled - unprofiled time per launch in
  // tools/kbench), 1.4 % of a 0.5 ms kernel.  With period k only every k-th launch of each kernel name is bracketed.
  int period = 1;
  std::vector<int> seen;             // launches met per slot (bracketed or not)
};

static hipEvent_t prof_event(pp_profiler* p) {
  if (!p->pool.empty()) {
    hipEvent_t e = p->pool.back();
    p->pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

void pp_prof_begin(pp_ctx* ctx, const char* kernel_name) {
  pp_profiler* p = ctx->prof;
  if (!p) return;
  int slot = -1;
  for (size_t i = 0; i < p->names.size(); ++i)
    if (p->names[i] == kernel_name) slot = (int)i;
  if (slot < 0) {
    p->names.emplace_back(kernel_name);
    slot = (int)p->names.size() - 1;
  }
  if ((int)p->seen.size() <= slot) p->seen.resize(slot + 1, 0);
  const int nth = p->seen[slot]++;
  // (the LAST launch of every group of `period` is the sampled one: launch 0 of a kernel name is atypical -- the first
  // iteration of a block reads the moving image itself -- and would otherwise be in every read-out)
  if (p->period > 1 && (nth % p->period) != p->period - 1) {   // not a sampled launch
    p->open_slot = -1;
    return;
  }
  p->open_slot = slot;
  p->open_a = prof_event(p);
  if (p->open_a) (void)hipEventRecord(p->open_a, ctx->stream);
}

void pp_prof_end(pp_ctx* ctx) {
  pp_profiler* p = ctx->prof;
  if (!p || p->open_slot < 0) return;
  hipEvent_t b = prof_event(p);
  if (b) (void)hipEventRecord(b, ctx->stream);
  if (p->open_a && b) p->spans.push_back({p->open_slot, p->open_a, b});
  p->open_slot = -1;
  p->open_a = nullptr;
}

int pp_fail(pp_ctx* ctx, int code, const char* fmt, ...) {
  if (ctx) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(ctx->err, sizeof(ctx->err), fmt, ap);
    va_end(ap);
  }
  return code;
}

// ---- the PP_* switches: one snapshot of the environment -----------------------------------------------------------------
namespace {
struct pp_switch_def {
  const char* name;
  bool numeric;   // the value must parse as an integer (else: presence is what counts)
};
constexpr pp_switch_def SWITCHES[] = {
    {"PP_CC_ROWS_BLOCK", false},  {"PP_COMPOSE_BLOCK", true},  {"PP_FIR_LEGACY", false},   {"PP_FIR_MARCH_SP", true},
    {"PP_FUSED_CUBE", true},      {"PP_FUSED_GEN", true},       {"PP_FUSED_MASK", true},     {"PP_FUSED_MIX", true},     {"PP_FUSED_NT", true},
    {"PP_FUSED_OPT", true},       {"PP_FUSED_PITCH", true},    {"PP_FUSED_SUM", true},     {"PP_FUSED_SYNC", true},
    {"PP_FUSED_TILE", true},      {"PP_FUSED_ZCHUNK", true},   {"PP_FUSED_ZCHUNK_A", true}, {"PP_FUSED_ZCHUNK_B", true},
    {"PP_GAUSS3", true},          {"PP_METRIC_BLOCKS", true},  {"PP_METRIC_GRAD_ONE_LAUNCH", true}, {"PP_METRIC_LANES", true},
    {"PP_METRIC_GRAD_PLANAR", false},
    {"PP_NO_FIXED_SAMPLES", false}, {"PP_POISON_WS", false},   {"PP_RESAMPLE_GENERIC", false}, {"PP_RG_GRID", true},
    {"PP_RG_SEG_V1", false},      {"PP_RG_TWO_SWEEP", false},  {"PP_RS_BAND", true},       {"PP_RS_ZCHUNK", true},
    {"PP_WARP_LEGACY", false},
};
constexpr int N_SWITCHES = (int)(sizeof(SWITCHES) / sizeof(SWITCHES[0]));
struct pp_switch_snapshot {
  bool has[N_SWITCHES];
  char value[N_SWITCHES][32];
};
std::atomic<pp_switch_snapshot*> g_switches{nullptr};
std::mutex g_switch_lock;

pp_switch_snapshot* take_snapshot() {
  auto* s = new pp_switch_snapshot();
  for (int i = 0; i < N_SWITCHES; ++i) {
    const char* v = getenv(SWITCHES[i].name);     // the ONLY getenv of the library: here, under g_switch_lock
    s->has[i] = false;
    s->value[i][0] = 0;
    if (!v) continue;
    if (SWITCHES[i].numeric) {
      char* end = nullptr;
      (void)strtol(v, &end, 10);
      if (end == v || *end != 0 || strlen(v) >= sizeof(s->value[i])) {
        fprintf(stderr, "platipy_amd: %s=\"%s\" is not an integer; the switch is ignored\n", SWITCHES[i].name, v);
        continue;
      }
    }
    s->has[i] = true;
    strncpy(s->value[i], v, sizeof(s->value[i]) - 1);
  }
  return s;
}
}  // namespace

const char* pp_env(const char* name) {
  pp_switch_snapshot* s = g_switches.load(std::memory_order_acquire);
  if (!s) {
    std::lock_guard<std::mutex> lock(g_switch_lock);
    s = g_switches.load(std::memory_order_acquire);
    if (!s) {
      s = take_snapshot();
      g_switches.store(s, std::memory_order_release);
    }
  }
  for (int i = 0; i < N_SWITCHES; ++i)
    if (strcmp(SWITCHES[i].name, name) == 0) return s->has[i] ? s->value[i] : nullptr;
  return nullptr;
}

// Re-read the environment (tests and A/B tools that flip a switch inside one process).  Not for use while another thread is
// inside the library: the previous snapshot is deliberately leaked, a launcher may still hold a pointer into it.
extern "C" void pp_reload_switches(void) {
  std::lock_guard<std::mutex> lock(g_switch_lock);
  g_switches.store(take_snapshot(), std::memory_order_release);
}

int pp_reserve(pp_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) {
    // debugging aid: PP_POISON_WS=1 fills the reserved scratch with NaN bytes on every call, so a kernel that reads
    // scratch it did not write shows up as NaN / changed results (tools/stress_streams.py)
    const bool poison = pp_env("PP_POISON_WS") != nullptr;
    if (poison && bytes) PP_HIP(ctx, hipMemsetAsync(ctx->ws, 0xFF, bytes, ctx->stream));
    return PP_OK;
  }
  // Growing is rare (first call at a size); wait for queued work that may still use ws.
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->ws) {
    PP_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
  }
  void* p = nullptr;
  const size_t want = pp_align_up(bytes, (size_t)1 << 20);
  if (hipMalloc(&p, want) != hipSuccess || !p) {
    (void)hipGetLastError();
    return pp_fail(ctx, PP_ERR_ALLOC, "workspace allocation of %zu bytes failed", want);
  }
  ctx->ws = static_cast<char*>(p);
  ctx->ws_bytes = want;
  return PP_OK;
}

int pp_read_back(pp_ctx* ctx, const void* dev, void* host, size_t bytes) {
  if (bytes > 4096) return pp_fail(ctx, PP_ERR_ARG, "pp_read_back: %zu bytes exceed the staging buffer", bytes);
  if (!ctx->pinned) {
    void* p = nullptr;
    if (hipHostMalloc(&p, 4096, 0) != hipSuccess || !p) {
      (void)hipGetLastError();
      return pp_fail(ctx, PP_ERR_ALLOC, "page-locked staging buffer allocation failed");
    }
    ctx->pinned = static_cast<char*>(p);
  }
  PP_HIP(ctx, hipMemcpyAsync(ctx->pinned, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  memcpy(host, ctx->pinned, bytes);
  return PP_OK;
}

int pp_mailbox(pp_ctx* ctx, char** mailbox, unsigned long long* seq) {
  if (!ctx->mailbox) {
    void* p = nullptr;
    if (hipHostMalloc(&p, PP_MAIL_WRITERS * PP_MAIL_SLOT, 0) != hipSuccess || !p) {
      (void)hipGetLastError();
      return pp_fail(ctx, PP_ERR_ALLOC, "mailbox allocation failed");
    }
    memset(p, 0, PP_MAIL_WRITERS * PP_MAIL_SLOT);
    ctx->mailbox = static_cast<char*>(p);
    ctx->mail_seq = 0;
  }
  *mailbox = ctx->mailbox;
  *seq = ++ctx->mail_seq;
  return PP_OK;
}

int pp_mail_take(pp_ctx* ctx, int writer, int n, unsigned long long seq, double* out) {
  if (writer < 0 || writer >= PP_MAIL_WRITERS || n < 0 || n > PP_MAIL_ENTRIES)
    return pp_fail(ctx, PP_ERR_ARG, "pp_mail_take: writer %d / %d entries exceed the mailbox", writer, n);
  volatile pp_mail_entry* e = pp_mail_slot(ctx->mailbox, writer);
  const auto t0 = std::chrono::steady_clock::now();
  bool synced = false;
  for (int k = 0; k < n; ++k) {
    for (unsigned spins = 0;; ++spins) {
      // an entry is taken when its two words verify each other for THIS launch (pp_mail_tag): a torn or stale pair does not
      const unsigned long long tag = e[k].tag;
      std::atomic_thread_fence(std::memory_order_acquire);
      const double value = e[k].value;
      if (tag == pp_mail_tag(seq, value)) {
        out[k] = value;
        break;
      }
      if ((spins & 1023u) != 1023u || std::chrono::steady_clock::now() - t0 <= std::chrono::milliseconds(200)) continue;
      // not there yet: wait for the stream the ordinary way (also surfaces a failed launch), then look once more
      if (synced) return pp_fail(ctx, PP_ERR_HIP, "kernel finished without posting its result");
      PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
      synced = true;
    }
  }
  return PP_OK;
}

int pp_ticket(pp_ctx* ctx, unsigned** out) {
  if (!ctx->ticket) {
    void* p = nullptr;
    if (hipMalloc(&p, 2048) != hipSuccess || !p) {   // 16 counters, 128 bytes apart
      (void)hipGetLastError();
      return pp_fail(ctx, PP_ERR_ALLOC, "ticket counter allocation failed");
    }
    ctx->ticket = static_cast<unsigned*>(p);
    PP_HIP(ctx, hipMemsetAsync(ctx->ticket, 0, 2048, ctx->stream));
  }
  *out = ctx->ticket;
  return PP_OK;
}

int pp_history_buffer(pp_ctx* ctx, double** out) {
  if (!ctx->hist) {
    void* p = nullptr;
    if (hipMalloc(&p, 2 * sizeof(double) * ((size_t)PP_HIST_CAP + 1)) != hipSuccess || !p) {
      (void)hipGetLastError();
      return pp_fail(ctx, PP_ERR_ALLOC, "iteration-history buffer allocation failed");
    }
    ctx->hist = static_cast<double*>(p);
    ctx->hist_cap = PP_HIST_CAP;
  }
  *out = ctx->hist;
  return PP_OK;
}

extern "C" {

int pp_abi_version(void) { return PP_ABI_VERSION; }

int pp_create(int device, void* hip_stream, pp_ctx** out) {
  if (!out) return PP_ERR_ARG;
  *out = nullptr;
  int prev = -1, count = 0;
  (void)hipGetDevice(&prev);
  if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {   // validated without touching the caller's device
    (void)hipGetLastError();
    return PP_ERR_HIP;
  }
  pp_ctx* c = static_cast<pp_ctx*>(calloc(1, sizeof(pp_ctx)));
  if (!c) return PP_ERR_ALLOC;
  c->device = device;
  c->stream = static_cast<hipStream_t>(hip_stream);
  *out = c;
  return PP_OK;
}

void pp_destroy(pp_ctx* ctx) {
  if (!ctx) return;
  pp_device_guard dev_guard_(ctx);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->ticket) (void)hipFree(ctx->ticket);
  if (ctx->hist) (void)hipFree(ctx->hist);
  if (ctx->fsamp) (void)hipFree(ctx->fsamp);
  if (ctx->pinned) (void)hipHostFree(ctx->pinned);
  if (ctx->mailbox) (void)hipHostFree(ctx->mailbox);
  if (ctx->prof) {
    for (auto& s : ctx->prof->spans) {
      (void)hipEventDestroy(s.a);
      (void)hipEventDestroy(s.b);
    }
    for (auto e : ctx->prof->pool) (void)hipEventDestroy(e);
    delete ctx->prof;
  }
  free(ctx);
}

int pp_profile_enable(pp_ctx* ctx, int on) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  if (on && !ctx->prof) ctx->prof = new pp_profiler();
  if (on && ctx->prof) ctx->prof->period = on > 1 ? on : 1;   // on = k > 1: bracket every k-th launch of each kernel
  if (!on && ctx->prof) {
    PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& s : ctx->prof->spans) {
      (void)hipEventDestroy(s.a);
      (void)hipEventDestroy(s.b);
    }
    for (auto e : ctx->prof->pool) (void)hipEventDestroy(e);
    delete ctx->prof;
    ctx->prof = nullptr;
  }
  return PP_OK;
}

int pp_profile_read(pp_ctx* ctx, pp_profile_entry* out, int cap) {
  if (!ctx || (!out && cap > 0)) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  pp_profiler* p = ctx->prof;
  if (!p) return 0;
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int n = (int)p->names.size();
  for (int i = 0; i < n && i < cap; ++i) {
    memset(&out[i], 0, sizeof(out[i]));
    snprintf(out[i].name, sizeof(out[i].name), "%s", p->names[i].c_str());
  }
  for (auto& s : p->spans) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess && s.slot < cap) {
      out[s.slot].launches += 1;
      out[s.slot].total_ms += (double)ms;
    }
    p->pool.push_back(s.a);
    p->pool.push_back(s.b);
  }
  p->spans.clear();
  std::fill(p->seen.begin(), p->seen.end(), 0);   // the sampling phase restarts with the accumulators
  return n < cap ? n : cap;
}

const char* pp_last_error(const pp_ctx* ctx) { return ctx ? ctx->err : "null ctx"; }

int pp_set_stream(pp_ctx* ctx, void* hip_stream) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  ctx->stream = static_cast<hipStream_t>(hip_stream);
  return PP_OK;
}

int pp_sync(pp_ctx* ctx) {
  if (!ctx) return PP_ERR_ARG;
  pp_device_guard dev_guard_(ctx);
  PP_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return PP_OK;
}

size_t pp_workspace_bytes(const pp_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

void pp_demons_default_params(pp_demons_params* p) {
  if (!p) return;
  // SimpleITK 2.3.1 FastSymmetricForcesDemonsRegistrationFilter defaults.
  p->iterations = 10;
  for (int k = 0; k < 3; ++k) {
    p->sigma_d_vox[k] = 1.0;
    p->sigma_u_vox[k] = 1.0;
  }
  p->smooth_displacement = 1;
  p->smooth_update = 0;
  p->max_rms_error = 0.02;
  p->max_step_length = 0.5;
  p->intensity_threshold = 0.001;
  p->denominator_threshold = 1e-9;
  p->max_error = 0.1;
  p->max_kernel_width = 30;
  p->variant = PP_DEMONS_AUTO;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Gaussian operator (host).  Discrete Gaussian kernel e^-t I_k(t) (Lindeberg), accumulated
// from the centre outwards until the covered mass reaches 1 - max_error, then normalised:
// the coefficient rule of itk::GaussianOperator, which every FIR on the reference's path
// uses (DiscreteGaussian, the demons field smoothers).  The modified Bessel functions use
// the classic Abramowitz-Stegun polynomial fits (9.8.1-9.8.4) and Miller's downward
// recurrence, as ITK does, so the taps agree with ITK's to double rounding.

namespace {

double bessel_i0(double x) {
  const double ax = std::fabs(x);
  if (ax < 3.75) {
    double y = x / 3.75;
    y *= y;
    return 1.0 + y * (3.5156229 + y * (3.0899424 + y * (1.2067492 + y * (0.2659732 + y * (0.360768e-1 + y * 0.45813e-2)))));
  }
  const double y = 3.75 / ax;
  return (std::exp(ax) / std::sqrt(ax)) *
         (0.39894228 + y * (0.1328592e-1 + y * (0.225319e-2 + y * (-0.157565e-2 + y * (0.916281e-2 +
          y * (-0.2057706e-1 + y * (0.2635537e-1 + y * (-0.1647633e-1 + y * 0.392377e-2))))))));
}

double bessel_i1(double x) {
  const double ax = std::fabs(x);
  double ans;
  if (ax < 3.75) {
    double y = x / 3.75;
    y *= y;
    ans = ax * (0.5 + y * (0.87890594 + y * (0.51498869 + y * (0.15084934 + y * (0.2658733e-1 +
          y * (0.301532e-2 + y * 0.32411e-3))))));
  } else {
    const double y = 3.75 / ax;
    ans = 0.2282967e-1 + y * (-0.2895312e-1 + y * (0.1787654e-1 - y * 0.420059e-2));
    ans = 0.39894228 + y * (-0.3988024e-1 + y * (-0.362018e-2 + y * (0.163801e-2 + y * (-0.1031555e-1 + y * ans))));
    ans *= std::exp(ax) / std::sqrt(ax);
  }
  return x < 0.0 ? -ans : ans;
}

double bessel_in(int n, double x) {
  if (x == 0.0) return 0.0;
  const double tox = 2.0 / std::fabs(x);
  double bip = 0.0, bi = 1.0, ans = 0.0;
  for (int j = 2 * (n + (int)(10.0 * std::sqrt((double)n))); j > 0; --j) {
    const double bim = bip + j * tox * bi;
    bip = bi;
    bi = bim;
    if (std::fabs(bi) > 1.0e10) {
      ans *= 1.0e-10;
      bi *= 1.0e-10;
      bip *= 1.0e-10;
    }
    if (j == n) ans = bip;
  }
  ans *= bessel_i0(x) / bi;
  return (x < 0.0 && (n & 1)) ? -ans : ans;
}

}  // namespace

extern "C" int pp_gauss_taps(double variance, double max_error, int max_kernel_width, double* taps, int cap) {
  if (!taps || cap < 3 || !(variance >= 0.0)) return PP_ERR_ARG;
  std::vector<double> half;
  const double et = std::exp(-variance);
  const double target = 1.0 - max_error;
  double mass = 0.0;
  half.push_back(et * bessel_i0(variance));
  mass += half[0];
  half.push_back(et * bessel_i1(variance));
  mass += 2.0 * half[1];
  for (int k = 2; mass < target; ++k) {
    half.push_back(et * bessel_in(k, variance));
    mass += 2.0 * half[k];
    if (half[k] < mass * DBL_EPSILON) break;              // no further mass can be gained
    if ((int)half.size() > max_kernel_width) break;       // width cap (one-sided count)
  }
  const int r = (int)half.size() - 1;
  if (2 * r + 1 > cap) return PP_ERR_SIZE;
  for (int k = 0; k <= r; ++k) {
    const double w = half[k] / mass;
    taps[r + k] = w;
    taps[r - k] = w;
  }
  return r;
}

int pp_make_taps(pp_ctx* ctx, double variance, double max_error, int max_kernel_width, pp_taps* t) {
  double w[2 * PP_MAX_RADIUS + 1];
  const int r = pp_gauss_taps(variance, max_error, max_kernel_width, w, 2 * PP_MAX_RADIUS + 1);
  if (r < 0) return pp_fail(ctx, r, "Gaussian operator (variance %g) needs more than %d taps", variance, 2 * PP_MAX_RADIUS + 1);
  t->r = r;
  for (int k = 0; k < 2 * r + 1; ++k) t->w[k] = (float)w[k];
  return PP_OK;
}

// ---------------------------------------------------------------------------------------
// geometry

int pp_geom_check(pp_ctx* ctx, const pp_geom* g, const char* what) {
  if (!g) return pp_fail(ctx, PP_ERR_ARG, "%s geometry is NULL", what);
  for (int k = 0; k < 3; ++k) {
    if (g->size[k] < 1) return pp_fail(ctx, PP_ERR_ARG, "%s size[%d] = %d", what, k, g->size[k]);
    if (!(g->spacing[k] > 0.0)) return pp_fail(ctx, PP_ERR_ARG, "%s spacing[%d] = %g", what, k, g->spacing[k]);
  }
  if ((double)g->size[0] * g->size[1] * g->size[2] >= 2147483647.0)
    return pp_fail(ctx, PP_ERR_SIZE, "%s has more than 2^31-1 voxels", what);
  return PP_OK;
}

bool pp_geom_identity_dir(const pp_geom* g) {
  static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k)
    if (g->direction[k] != I[k]) return false;
  return true;
}

bool pp_geom_same_grid(const pp_geom* a, const pp_geom* b) {
  for (int k = 0; k < 3; ++k)
    if (a->size[k] != b->size[k] || a->spacing[k] != b->spacing[k] || a->origin[k] != b->origin[k]) return false;
  for (int k = 0; k < 9; ++k)
    if (a->direction[k] != b->direction[k]) return false;
  return true;
}

static void mat3_mul(const double* a, const double* b, double* c) {
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) c[r * 3 + k] = a[r * 3 + 0] * b[0 * 3 + k] + a[r * 3 + 1] * b[1 * 3 + k] + a[r * 3 + 2] * b[2 * 3 + k];
}

static void mat3_inv(const double* m, double* r) {
  const double a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double id = 1.0 / (a * A + b * B + c * C);
  r[0] = A * id;
  r[1] = -(b * i - c * h) * id;
  r[2] = (b * f - c * e) * id;
  r[3] = B * id;
  r[4] = (a * i - c * g) * id;
  r[5] = -(a * f - c * d) * id;
  r[6] = C * id;
  r[7] = -(a * h - b * g) * id;
  r[8] = (a * e - b * d) * id;
}

void pp_make_index_map(const pp_geom* gin, const pp_geom* gout, const double* affine_A, const double* affine_t,
                       pp_index_map* m) {
  double i2p_out[9], i2p_in[9], p2i_in[9];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      i2p_out[r * 3 + c] = gout->direction[r * 3 + c] * gout->spacing[c];
      i2p_in[r * 3 + c] = gin->direction[r * 3 + c] * gin->spacing[c];
    }
  if (pp_geom_identity_dir(gin)) {
    memset(p2i_in, 0, sizeof(p2i_in));
    for (int k = 0; k < 3; ++k) p2i_in[k * 3 + k] = 1.0 / gin->spacing[k];
  } else {
    mat3_inv(i2p_in, p2i_in);
  }
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  static const double Z3[3] = {0, 0, 0};
  const double* XA = affine_A ? affine_A : I3;
  const double* Xt = affine_t ? affine_t : Z3;
  // c = P2I_in * (XA * (I2P_out * idx + o_out) + Xt - o_in)
  double t1[9];
  mat3_mul(XA, i2p_out, t1);
  mat3_mul(p2i_in, t1, m->A);
  double v[3];
  for (int r = 0; r < 3; ++r)
    v[r] = XA[r * 3 + 0] * gout->origin[0] + XA[r * 3 + 1] * gout->origin[1] + XA[r * 3 + 2] * gout->origin[2] + Xt[r] - gin->origin[r];
  for (int r = 0; r < 3; ++r) m->b[r] = p2i_in[r * 3 + 0] * v[0] + p2i_in[r * 3 + 1] * v[1] + p2i_in[r * 3 + 2] * v[2];
  memcpy(m->Md, p2i_in, sizeof(p2i_in));
}
