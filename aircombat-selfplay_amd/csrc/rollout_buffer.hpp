// Device-resident rollout buffer (include/aircombat_buffer.h): the reference's ReplayBuffer /
// SharedReplayBuffer (algorithms/utils/buffer.py) with every array in HBM. Included at the end of aircombat.hip (same library,
// same error channel).
//
// Kernels, all HBM-bound streaming work (no reuse, no contraction):
//   returns_kernel<GAE, PROPER>   the reverse-time recurrence of compute_returns (buffer.py:134-167): one lane per (env, agent)
//                                 column, float32 operations in the reference's order with fused multiply-add contraction switched off, so
//                                 the result is bit-identical to the numpy code; the (independent) loads are spread over eight waves
//                                 per 64 columns and staged through LDS one tile ahead, so the dependent chain only touches LDS
//   advantage_* kernels           returns - values, fp64 mean / variance by block reduction, normalisation (buffer.py:72-75)
//   gather_rows_kernel            a mini-batch of recurrent_generator (buffer.py:196-268): rows of the column-major sequence view
#pragma once

struct ac_buffer {
  ac_buffer_config_t cfg;
  int T, N, device;
  int step;
  hipStream_t stream;
  float* f[AC_BUF_NFIELDS];
  int dim[AC_BUF_NFIELDS];      // floats per (t, column)
  int slots[AC_BUF_NFIELDS];    // T or T + 1
  double* d_stat;               // [2]: sum, sum of squared deviations
  int32_t* d_chunks; int chunks_cap;
  float* d_stage; int64_t stage_cap;     // device staging for host-side mini-batch outputs
  hipEvent_t ev0, ev1;
  float last_ms;
};

namespace rbuf {
// One workgroup = COLS columns x 512 threads. Only the recurrence is sequential: the loads are not, so all eight waves fetch a tile
// of TT time steps x COLS columns of the four input arrays (32 loads in flight per lane, issued one tile AHEAD), stage it in LDS, the
// first COLS lanes run the TT dependent steps out of LDS while the next tile's loads fly, and all waves store the tile of returns.
// 64 KB in flight per workgroup. COLS = 32 (128-byte rows, TT = 128) doubles the workgroup count when 64-column workgroups would
// not cover the 256 CUs: the BASELINE batch's 8192 columns become 256 workgroups with 16 MB in flight.
// The recurrence runs on a ninth wave that issues no global loads of its own: a wave that both prefetches and scans would have
// to drain its own prefetch (in-order vmcnt) before the first dependent operation of the scan.
constexpr int NT = 512, NT_ALL = NT + 64;
template <bool GAE, bool PROPER, int COLS, int TT>
__global__ __launch_bounds__(NT_ALL) void returns_kernel(const float* __restrict__ r, const float* __restrict__ v, const float* __restrict__ m,
                                                     const float* __restrict__ b, float* __restrict__ R, int T, int N, float g, float gl) {
#pragma clang fp contract(off)   // every product is rounded before it is added, as in the numpy code (plain operators on purpose:
                                 // HIP's __fmul_rn / __fadd_rn are header inlines compiled with contraction allowed)
  constexpr int SPW = TT / (NT / COLS);   // time steps per thread and tile
  __shared__ float sr[TT][COLS], sm[TT][COLS], sb[PROPER ? TT : 1][COLS], sv[(GAE || PROPER) ? TT : 1][COLS], sR[TT][COLS];
  const bool scanner = threadIdx.x >= NT;
  const int lane = threadIdx.x % COLS, w = (threadIdx.x % NT) / COLS;
  const int n = blockIdx.x * COLS + lane;
  const bool col_ok = n < N;
  const size_t sN = (size_t)N, col = (size_t)(col_ok ? n : 0);
  float pr[SPW], pm[SPW], pb[SPW], pv[SPW], pn[SPW];
  auto fetch = [&](int t_hi) {
#pragma unroll
    for (int u = 0; u < SPW; ++u) {
      // rows before t = 0 (ragged last tile) and columns past N (ragged last workgroup) read a valid clamped address instead of
      // branching; what they read is never used: the recurrence stops at t = 0 and the stores are guarded
      const int t = max(t_hi - (w * SPW + u), 0);
      const size_t at = (size_t)t * sN + col, at1 = at + sN;
      pr[u] = r[at];
      pm[u] = m[at1];
      if (PROPER) pb[u] = b[at1];
      if (GAE || PROPER) pv[u] = v[at];
      if (GAE) pn[u] = v[at1];      // V[t+1] (the neighbouring step's V[t]: the same cache line, an L2 hit)
    }
  };
  // carried by the scanning lanes across steps: the value at t+1 (GAE: V[t+1]; MC: R[t+1]) and the accumulator
  float next = (scanner && !GAE) ? R[(size_t)T * sN + col] : 0.0f;
  float acc = 0.0f;
  if (!scanner) fetch(T - 1);
  for (int t_hi = T - 1; t_hi >= 0; t_hi -= TT) {
    if (!scanner) {
#pragma unroll
      for (int u = 0; u < SPW; ++u) {
        const int s_ = w * SPW + u;
        // GAE: the TD residual and the decay factor do not depend on the accumulator, so the loading waves compute them (same
        // operations, same order as the reference's expression) and only  acc = delta + c * acc  is left on the dependent chain
        sr[s_][lane] = GAE ? (pr[u] + (g * pn[u]) * pm[u]) - pv[u] : pr[u];
        sm[s_][lane] = GAE ? gl * pm[u] : pm[u];
        if (PROPER) sb[s_][lane] = pb[u];
        if (GAE || PROPER) sv[s_][lane] = pv[u];
      }
    }
    __syncthreads();
    if (!scanner && t_hi - TT >= 0) fetch(t_hi - TT);       // the next tile's loads fly while the ninth wave runs this tile's recurrence
    if (scanner && threadIdx.x - NT < COLS) {
      const int steps = min(TT, t_hi + 1);
#pragma unroll 8
      for (int s_ = 0; s_ < steps; ++s_) {
        const float rt = sr[s_][lane], mt = sm[s_][lane];
        float out;
        if (GAE) {
          acc = rt + mt * acc;                       // rt = delta_t, mt = gamma * lambda * mask_{t+1}
          if (PROPER) acc = acc * sb[s_][lane];
          out = acc + sv[s_][lane];
        } else {
          const float disc = (next * g) * mt + rt;
          out = PROPER ? disc * sb[s_][lane] + (1.0f - sb[s_][lane]) * sv[s_][lane] : disc;
          next = out;
        }
        sR[s_][lane] = out;
      }
    }
    __syncthreads();
    if (!scanner) {
#pragma unroll
      for (int u = 0; u < SPW; ++u) {
        const int s_ = w * SPW + u, t = t_hi - s_;
        if (t >= 0 && col_ok) R[(size_t)t * sN + col] = sR[s_][lane];
      }
    }
  }
}

__device__ __forceinline__ double block_sum(double x) {
  __shared__ double part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = x;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0) for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += part[w];
  __syncthreads();
  return s;   // valid in thread 0
}
// pass 0: adv = R - V, sum; pass 1: sum of squared deviations from the mean
__global__ __launch_bounds__(256) void advantage_stats_kernel(const float* __restrict__ R, const float* __restrict__ V, float* __restrict__ adv,
                                                              int64_t count, double* stat, int pass) {
  double s = 0.0;
  const double mean = pass ? stat[0] / (double)count : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    if (pass == 0) { const float a = R[i] - V[i]; adv[i] = a; s += (double)a; }
    else { const double d = (double)adv[i] - mean; s += d * d; }
  }
  s = block_sum(s);
  if (threadIdx.x == 0) atomicAdd(&stat[pass], s);
}
__global__ __launch_bounds__(256) void advantage_normalise_kernel(float* __restrict__ adv, int64_t count, const double* stat) {
  const float mean = (float)(stat[0] / (double)count);
  const float sd = (float)sqrt(stat[1] / (double)count);
  const float den = sd + 1e-5f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    adv[i] = (adv[i] - mean) / den;
}

// out[(l * n_chunks + j) * dim + k] = src[(t * N + col) * dim + k], row = chunks[j] * L + l, col = row / T, t = row % T
// (first_only: the RNN states, one row per chunk: l = 0 only)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, float* __restrict__ out, const int32_t* __restrict__ chunks,
                                                          int n_chunks, int L, int T, int N, int dim, int first_only) {
  const int64_t total = (int64_t)(first_only ? 1 : L) * n_chunks * dim;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % dim);
    const int64_t rowi = i / dim;
    const int j = (int)(rowi % n_chunks), l = (int)(rowi / n_chunks);
    const int64_t row = (int64_t)chunks[j] * L + l;
    const int col = (int)(row / T), t = (int)(row % T);
    out[i] = src[((int64_t)t * N + col) * dim + k];
  }
}
__global__ void fill_kernel(float* p, int64_t n, float v) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) p[i] = v;
}
}  // namespace rbuf

static int64_t buf_count(const ac_buffer* b, int fld) { return (int64_t)b->slots[fld] * b->N * b->dim[fld]; }
static int buf_reset_arrays(ac_buffer* b) {
  for (int k = 0; k < AC_BUF_NFIELDS; ++k) {
    if (!b->f[k]) continue;
    const bool ones = (k == AC_BUF_MASKS || k == AC_BUF_BAD_MASKS || k == AC_BUF_ACTIVE_MASKS);
    const int64_t n = buf_count(b, k);
    if (ones) hipLaunchKernelGGL(rbuf::fill_kernel, dim3(1024), dim3(256), 0, b->stream, b->f[k], n, 1.0f);
    else HIP_OK(hipMemsetAsync(b->f[k], 0, (size_t)n * sizeof(float), b->stream));
  }
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(b->stream));
  b->step = 0;
  return 0;
}

extern "C" {

ac_buffer_t* ac_buffer_create(const ac_buffer_config_t* cfg, int device_id) {
  auto bad = [](const char* m) -> ac_buffer_t* { fail(m); return nullptr; };
  if (!cfg) return bad("ac_buffer_create: null config");
  if (cfg->buffer_size <= 0 || cfg->n_envs <= 0 || cfg->n_agents <= 0 || cfg->obs_dim <= 0 || cfg->act_dim <= 0 || cfg->logp_dim <= 0 ||
      cfg->share_obs_dim < 0 || cfg->hidden_layers <= 0 || cfg->hidden_size <= 0)
    return bad("ac_buffer_create: sizes must be positive (share_obs_dim may be 0)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return bad("ac_buffer_create: no such HIP device");
  if (hipSetDevice(device_id) != hipSuccess) return bad("ac_buffer_create: hipSetDevice failed");
  ac_buffer* b = new ac_buffer();
  memset(b, 0, sizeof *b);
  b->cfg = *cfg; b->T = cfg->buffer_size; b->N = cfg->n_envs * cfg->n_agents; b->device = device_id;
  const int T = b->T, hid = cfg->hidden_layers * cfg->hidden_size;
  const int dims[AC_BUF_NFIELDS] = {cfg->obs_dim, cfg->share_obs_dim, cfg->act_dim, 1, 1, 1, cfg->share_obs_dim > 0 ? 1 : 0, cfg->logp_dim, 1, 1, hid, hid, 1};
  const int slots[AC_BUF_NFIELDS] = {T + 1, T + 1, T, T, T + 1, T + 1, T + 1, T, T + 1, T + 1, T + 1, T + 1, T};
  bool ok = hipStreamCreate(&b->stream) == hipSuccess && hipEventCreate(&b->ev0) == hipSuccess && hipEventCreate(&b->ev1) == hipSuccess;
  for (int k = 0; k < AC_BUF_NFIELDS && ok; ++k) {
    b->dim[k] = dims[k]; b->slots[k] = slots[k];
    if (dims[k] > 0) ok = hipMalloc(&b->f[k], (size_t)buf_count(b, k) * sizeof(float)) == hipSuccess;
  }
  ok = ok && hipMalloc(&b->d_stat, 2 * sizeof(double)) == hipSuccess;
  if (!ok || buf_reset_arrays(b) != 0) { ac_buffer_destroy(b); return bad("ac_buffer_create: device allocation failed"); }
  return b;
}

void ac_buffer_destroy(ac_buffer_t* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (int k = 0; k < AC_BUF_NFIELDS; ++k) if (b->f[k]) (void)hipFree(b->f[k]);
  if (b->d_stat) (void)hipFree(b->d_stat);
  if (b->d_chunks) (void)hipFree(b->d_chunks);
  if (b->d_stage) (void)hipFree(b->d_stage);
  if (b->ev0) (void)hipEventDestroy(b->ev0);
  if (b->ev1) (void)hipEventDestroy(b->ev1);
  if (b->stream) (void)hipStreamDestroy(b->stream);
  delete b;
}

int ac_buffer_step_index(const ac_buffer_t* b) { return b ? b->step : -1; }

static int buf_put(ac_buffer* b, int fld, int t, const float* src, int on_device) {
  if (!src) return 0;
  if (!b->f[fld]) return fail("ac_buffer: this buffer has no such field (share_obs / active_masks need share_obs_dim > 0)");
  const size_t n = (size_t)b->N * b->dim[fld];
  HIP_OK(hipMemcpyAsync(b->f[fld] + (size_t)t * n, src, n * sizeof(float), on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, b->stream));
  return 0;
}

int ac_buffer_insert(ac_buffer_t* b, const ac_buffer_step_t* s, int on_device) {
  if (!b || !s) return fail("ac_buffer_insert: null argument");
  if (!s->obs || !s->actions || !s->rewards || !s->masks || !s->action_log_probs || !s->value_preds || !s->rnn_states_actor || !s->rnn_states_critic)
    return fail("ac_buffer_insert: obs, actions, rewards, masks, action_log_probs, value_preds and both rnn states are required");
  HIP_OK(hipSetDevice(b->device));
  const int t = b->step;
  const bool shared = b->cfg.share_obs_dim > 0;
  if (shared && !s->share_obs) return fail("ac_buffer_insert: share_obs is required by the shared buffer");
  int rc = 0;
  rc |= buf_put(b, AC_BUF_OBS, t + 1, s->obs, on_device);
  rc |= buf_put(b, AC_BUF_MASKS, t + 1, s->masks, on_device);
  rc |= buf_put(b, AC_BUF_RNN_ACTOR, t + 1, s->rnn_states_actor, on_device);
  rc |= buf_put(b, AC_BUF_RNN_CRITIC, t + 1, s->rnn_states_critic, on_device);
  rc |= buf_put(b, AC_BUF_ACTIONS, t, s->actions, on_device);
  rc |= buf_put(b, AC_BUF_REWARDS, t, s->rewards, on_device);
  rc |= buf_put(b, AC_BUF_LOGP, t, s->action_log_probs, on_device);
  rc |= buf_put(b, AC_BUF_VALUES, t, s->value_preds, on_device);
  if (shared) {   // buffer.py:338-343: share_obs and active_masks are stored, bad_masks is not forwarded to the base class
    rc |= buf_put(b, AC_BUF_SHARE_OBS, t + 1, s->share_obs, on_device);
    rc |= buf_put(b, AC_BUF_ACTIVE_MASKS, t + 1, s->active_masks, on_device);
  } else {
    rc |= buf_put(b, AC_BUF_BAD_MASKS, t + 1, s->bad_masks, on_device);
  }
  if (rc) return -1;
  HIP_OK(hipStreamSynchronize(b->stream));   // host sources may be reused by the caller right away
  b->step = (t + 1) % b->T;
  return 0;
}

int ac_buffer_after_update(ac_buffer_t* b) {
  if (!b) return fail("ac_buffer_after_update: null handle");
  HIP_OK(hipSetDevice(b->device));
  const int rolled[] = {AC_BUF_OBS, AC_BUF_MASKS, AC_BUF_BAD_MASKS, AC_BUF_RNN_ACTOR, AC_BUF_RNN_CRITIC, AC_BUF_ACTIVE_MASKS, AC_BUF_SHARE_OBS};
  for (int k : rolled) {
    if (!b->f[k]) continue;
    const size_t n = (size_t)b->N * b->dim[k];
    HIP_OK(hipMemcpyAsync(b->f[k], b->f[k] + (size_t)b->T * n, n * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
  }
  HIP_OK(hipStreamSynchronize(b->stream));
  return 0;
}

int ac_buffer_clear(ac_buffer_t* b) {
  if (!b) return fail("ac_buffer_clear: null handle");
  HIP_OK(hipSetDevice(b->device));
  return buf_reset_arrays(b);
}

int ac_buffer_compute_returns(ac_buffer_t* b, const float* next_value, int on_device) {
  if (!b || !next_value) return fail("ac_buffer_compute_returns: null argument");
  HIP_OK(hipSetDevice(b->device));
  const int T = b->T, N = b->N;
  const bool gae = b->cfg.use_gae, proper = b->cfg.use_proper_time_limits;
  // value_preds[-1] = next_value (GAE) / returns[-1] = next_value (buffer.py:143,151,158,165)
  if (buf_put(b, gae ? AC_BUF_VALUES : AC_BUF_RETURNS, T, next_value, on_device)) return -1;
  const float g = (float)b->cfg.gamma, gl = (float)(b->cfg.gamma * b->cfg.gae_lambda);
  const float *r = b->f[AC_BUF_REWARDS], *v = b->f[AC_BUF_VALUES], *m = b->f[AC_BUF_MASKS], *bm = b->f[AC_BUF_BAD_MASKS];
  float* R = b->f[AC_BUF_RETURNS];
  const bool wide = N >= 64 * 256;   // enough 64-column workgroups for every CU
  dim3 grid(wide ? (N + 63) / 64 : (N + 31) / 32), block(rbuf::NT_ALL);
  HIP_OK(hipEventRecord(b->ev0, b->stream));
#define AC_RET(G, P)                                                                                                                \
  do {                                                                                                                              \
    if (wide) hipLaunchKernelGGL((rbuf::returns_kernel<G, P, 64, 64>), grid, block, 0, b->stream, r, v, m, bm, R, T, N, g, gl);     \
    else hipLaunchKernelGGL((rbuf::returns_kernel<G, P, 32, 128>), grid, block, 0, b->stream, r, v, m, bm, R, T, N, g, gl);        \
  } while (0)
  if (gae && proper) AC_RET(true, true); else if (gae) AC_RET(true, false); else if (proper) AC_RET(false, true); else AC_RET(false, false);
#undef AC_RET
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(b->ev1, b->stream));
  HIP_OK(hipStreamSynchronize(b->stream));
  HIP_OK(hipEventElapsedTime(&b->last_ms, b->ev0, b->ev1));
  return 0;
}

int ac_buffer_last_kernel_ms(ac_buffer_t* b, float* ms) {
  if (!b || !ms) return fail("ac_buffer_last_kernel_ms: null argument");
  *ms = b->last_ms;
  return 0;
}

int ac_buffer_advantages(ac_buffer_t* b) {
  if (!b) return fail("ac_buffer_advantages: null handle");
  HIP_OK(hipSetDevice(b->device));
  const int64_t count = (int64_t)b->T * b->N;
  HIP_OK(hipMemsetAsync(b->d_stat, 0, 2 * sizeof(double), b->stream));
  const int blocks = (int)std::min<int64_t>(2048, (count + 255) / 256);
  float* adv = b->f[AC_BUF_ADVANTAGES];
  hipLaunchKernelGGL(rbuf::advantage_stats_kernel, dim3(blocks), dim3(256), 0, b->stream, b->f[AC_BUF_RETURNS], b->f[AC_BUF_VALUES], adv, count, b->d_stat, 0);
  hipLaunchKernelGGL(rbuf::advantage_stats_kernel, dim3(blocks), dim3(256), 0, b->stream, b->f[AC_BUF_RETURNS], b->f[AC_BUF_VALUES], adv, count, b->d_stat, 1);
  hipLaunchKernelGGL(rbuf::advantage_normalise_kernel, dim3(blocks), dim3(256), 0, b->stream, adv, count, b->d_stat);
  HIP_OK(hipGetLastError());
  HIP_OK(hipStreamSynchronize(b->stream));
  return 0;
}

int ac_buffer_minibatch(ac_buffer_t* b, const int32_t* chunks, int32_t n_chunks, int32_t L, const ac_buffer_batch_t* out, int on_device) {
  if (!b || !chunks || !out) return fail("ac_buffer_minibatch: null argument");
  if (n_chunks <= 0 || L <= 0) return fail("ac_buffer_minibatch: n_chunks and chunk_len must be positive");
  const int64_t rows = (int64_t)b->T * b->N;
  for (int j = 0; j < n_chunks; ++j)
    if (chunks[j] < 0 || ((int64_t)chunks[j] + 1) * L > rows) return fail("ac_buffer_minibatch: chunk index outside the buffer");
  HIP_OK(hipSetDevice(b->device));
  if (n_chunks > b->chunks_cap) {
    if (b->d_chunks) HIP_OK(hipFree(b->d_chunks));
    HIP_OK(hipMalloc(&b->d_chunks, (size_t)n_chunks * sizeof(int32_t)));
    b->chunks_cap = n_chunks;
  }
  HIP_OK(hipMemcpyAsync(b->d_chunks, chunks, (size_t)n_chunks * sizeof(int32_t), hipMemcpyHostToDevice, b->stream));
  struct Item { int fld; float* dst; int first_only; };
  const Item items[] = {{AC_BUF_OBS, out->obs, 0}, {AC_BUF_SHARE_OBS, out->share_obs, 0}, {AC_BUF_ACTIONS, out->actions, 0}, {AC_BUF_MASKS, out->masks, 0},
                        {AC_BUF_ACTIVE_MASKS, out->active_masks, 0}, {AC_BUF_LOGP, out->action_log_probs, 0}, {AC_BUF_ADVANTAGES, out->advantages, 0},
                        {AC_BUF_RETURNS, out->returns, 0}, {AC_BUF_VALUES, out->value_preds, 0},
                        {AC_BUF_RNN_ACTOR, out->rnn_states_actor, 1}, {AC_BUF_RNN_CRITIC, out->rnn_states_critic, 1}};
  for (const Item& it : items) {
    if (!it.dst) continue;
    if (!b->f[it.fld]) return fail("ac_buffer_minibatch: this buffer has no such field");
    const int dim = b->dim[it.fld];
    const int64_t total = (int64_t)(it.first_only ? 1 : L) * n_chunks * dim;
    float* dev_dst = it.dst;
    if (!on_device) {
      if (total > b->stage_cap) {
        if (b->d_stage) HIP_OK(hipFree(b->d_stage));
        HIP_OK(hipMalloc(&b->d_stage, (size_t)total * sizeof(float)));
        b->stage_cap = total;
      }
      dev_dst = b->d_stage;
    }
    const int blocks = (int)std::min<int64_t>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(rbuf::gather_rows_kernel, dim3(blocks), dim3(256), 0, b->stream, b->f[it.fld], dev_dst, b->d_chunks, n_chunks, L, b->T, b->N, dim, it.first_only);
    HIP_OK(hipGetLastError());
    if (!on_device) {
      HIP_OK(hipMemcpyAsync(it.dst, dev_dst, (size_t)total * sizeof(float), hipMemcpyDeviceToHost, b->stream));
      HIP_OK(hipStreamSynchronize(b->stream));   // the staging buffer is reused by the next field
    }
  }
  HIP_OK(hipStreamSynchronize(b->stream));
  return 0;
}

int ac_buffer_device_ptr(ac_buffer_t* b, int32_t field, float** ptr, int64_t* n_floats) {
  if (!b || !ptr || field < 0 || field >= AC_BUF_NFIELDS) return fail("ac_buffer_device_ptr: bad argument");
  if (!b->f[field]) return fail("ac_buffer_device_ptr: this buffer has no such field");
  *ptr = b->f[field];
  if (n_floats) *n_floats = buf_count(b, field);
  return 0;
}

int ac_buffer_read(ac_buffer_t* b, int32_t field, float* host_out) {
  if (!b || !host_out || field < 0 || field >= AC_BUF_NFIELDS) return fail("ac_buffer_read: bad argument");
  if (!b->f[field]) return fail("ac_buffer_read: this buffer has no such field");
  HIP_OK(hipSetDevice(b->device));
  HIP_OK(hipMemcpy(host_out, b->f[field], (size_t)buf_count(b, field) * sizeof(float), hipMemcpyDeviceToHost));
  return 0;
}

int ac_buffer_write_slot(ac_buffer_t* b, int32_t field, int32_t t, const float* host_in) {
  if (!b || !host_in || field < 0 || field >= AC_BUF_NFIELDS) return fail("ac_buffer_write_slot: bad argument");
  if (!b->f[field]) return fail("ac_buffer_write_slot: this buffer has no such field");
  if (t < 0 || t >= b->slots[field]) return fail("ac_buffer_write_slot: time slot out of range");
  HIP_OK(hipSetDevice(b->device));
  if (buf_put(b, field, t, host_in, 0)) return -1;
  HIP_OK(hipStreamSynchronize(b->stream));
  return 0;
}

}  // extern "C"
