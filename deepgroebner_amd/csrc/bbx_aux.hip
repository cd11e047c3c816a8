// Housekeeping kernels of libbbx (initialisation, queue refill, ragged observations, clones, record growth, header
// gathers), the stand-alone PMLP policy launchers and the launcher that picks a kernel class (bbx_launch_step).
#include "bbx_device.h"
#include "bbx_pmlp.h"
#include "bbx_binom.h"

extern "C" int bbx_launch_general(const BbxParams* p, int kind, int blocks, int threads, size_t lds, hipStream_t stream);
extern "C" int bbx_launch_binom(const BbxParams* p, int kind, int blocks, int threads, size_t lds, hipStream_t stream);
extern "C" int bbx_launch_fast(const BbxParams* p, int blocks, int threads, int envs_per_block, hipStream_t stream);
extern "C" int bbx_launch_wide(const BbxParams* p, int nw, hipStream_t stream);

// ------------------------------------------------------------------ housekeeping kernels
// zero the headers and set the per-environment agent seeds
__global__ void bbx_init_kernel(char* recs, uint32_t rec_bytes, int B, const uint32_t* agent_seeds) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  BbxHdr* h = (BbxHdr*)(recs + (size_t)env * rec_bytes);
  BbxHdr z = {};
  z.agent_seed = agent_seeds ? agent_seeds[env] : (uint32_t)env;
  z.std_rng = 1u;                               // std::default_random_engine's default seed
  *h = z;
}
// request a reset (mask == null: every environment); clears a sticky error so the slot can be reused
__global__ void bbx_mark_reset_kernel(char* recs, uint32_t rec_bytes, int B, const uint8_t* mask) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  if (mask && !mask[env]) return;
  BbxHdr* h = (BbxHdr*)(recs + (size_t)env * rec_bytes);
  h->need_reset = 1; h->status = BBX_ST_OK; h->nP = 0; h->nG = 0; h->arena_used = 0; h->done_last = 0;
}
// refill of the ideal queue: the host stages the rings of the environments it topped up ([n ids][n tails][n rings]) and
// uploads them with one copy; this kernel moves every ring to its place (one workgroup per ring)
__global__ void bbx_scatter_queue_kernel(const uint32_t* stage, int n, uint32_t ring_words, uint32_t* q, int32_t* tail) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const int env = (int)stage[i];
  const uint32_t* src = stage + 2 * (size_t)n + (size_t)i * ring_words;
  uint32_t* dst = q + (size_t)env * ring_words;
  for (uint32_t w = threadIdx.x; w < ring_words; w += blockDim.x) dst[w] = src[w];
  if (threadIdx.x == 0) tail[env] = (int32_t)stage[n + i];
}
extern "C" int bbx_launch_scatter_queue(const uint32_t* stage, int n, uint32_t ring_words, uint32_t* q, int32_t* tail, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_scatter_queue_kernel, dim3(n), dim3(256), 0, stream, stage, n, ring_words, q, tail);
  return (int)hipGetLastError();
}
// ragged observation: the rows of every environment back to back (what a list of per-environment matrices needs),
// packed on the device from the padded block a step launch leaves behind.  Kernel 1: off[e] = sum of min(rows, cap)
// over the environments before e (one workgroup); kernel 2: one workgroup per environment copies its rows.
__global__ __launch_bounds__(1024) void bbx_obs_offsets_kernel(const int32_t* rows, int B, int cap, int32_t* off) {
  __shared__ int part[1024];
  const int t = threadIdx.x, per = (B + 1023) / 1024;
  int s = 0;
  for (int i = 0; i < per; i++) { const int e = t * per + i; if (e < B) { const int r = rows[e]; s += r < cap ? r : cap; } }
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {                     // inclusive scan (Hillis-Steele)
    const int v = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  int base = t ? part[t - 1] : 0;
  for (int i = 0; i < per; i++) {
    const int e = t * per + i;
    if (e < B) { off[e] = base; const int r = rows[e]; base += r < cap ? r : cap; }
  }
  if (t == 1023) off[B] = part[1023];
}
__global__ void bbx_obs_pack_kernel(const int32_t* padded, int cap, int cols, const int32_t* off, int B, int32_t* packed) {
  const int e = blockIdx.x;
  if (e >= B) return;
  const int n = (off[e + 1] - off[e]) * cols;
  const int32_t* src = padded + (size_t)e * cap * cols;
  int32_t* dst = packed + (size_t)off[e] * cols;
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}
extern "C" int bbx_launch_obs_pack(const int32_t* padded, int cap, int cols, const int32_t* rows, int B, int32_t* off, int32_t* packed, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_obs_offsets_kernel, dim3(1), dim3(1024), 0, stream, rows, B, cap, off);
  hipLaunchKernelGGL(bbx_obs_pack_kernel, dim3(B), dim3(64), 0, stream, padded, cap, cols, off, B, packed);
  return (int)hipGetLastError();
}
extern "C" int bbx_launch_init(char* recs, uint32_t rec_bytes, int B, const uint32_t* agent_seeds, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_init_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, recs, rec_bytes, B, agent_seeds);
  return (int)hipGetLastError();
}
extern "C" int bbx_launch_mark_reset(char* recs, uint32_t rec_bytes, int B, const uint8_t* mask, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_mark_reset_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, recs, rec_bytes, B, mask);
  return (int)hipGetLastError();
}

// value(): clone environment src[k] of one record array into slot k of another (live prefixes only), optionally
// re-seeding the built-in random agent of the clone
// seed_std != 0: the seeds are states of the std::default_random_engine behind the seeded Random selection (BbxHdr.std_rng)
// instead of seeds of the counter-hash agent.  flags (or null): per clone, bit 0 = the source is in an error state, bit 1 =
// two of its first ngen basis elements (the ideal's generators) have the same lead monomial (value(): buchberger() sorts
// its reducers with std::sort, whose order of equal elements the host reproduces — bbx_api.cpp).
template <int W>
__global__ void bbx_clone_kernel(const char* src_recs, char* dst_recs, BbxLayout L, const int32_t* src, const int32_t* dst, int n,
                                 const uint32_t* seeds, int keep_counters, int seed_std, int ngen, uint8_t* flags) {
  const int k = blockIdx.x * (blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
  if (k >= n) return;
  char* s = const_cast<char*>(src_recs) + (size_t)src[k] * L.rec_bytes;
  char* d = dst_recs + (size_t)(dst ? dst[k] : k) * L.rec_bytes;
  BbxHdr h = *(const BbxHdr*)s;
  if (L.kind == 1) bstage_copy<W>(benv_view<W>(d, L), benv_view<W>(s, L), h.nG, h.nP);
  else stage_copy<W>(env_view<W>(d, L), env_view<W>(s, L), h.nG, h.nP, h.arena_used);
  if (flags) {
    const Mono<W>* lm = (const Mono<W>*)(s + L.off_lm);
    const int ng = h.nG < ngen ? h.nG : ngen;
    bool tie = false;
    for (int i = lane_id(); i < ng; i += WAVE) {
      const Mono<W> mi = lm[i];
      for (int j = i + 1; j < ng; j++) tie = tie || m_eq(mi, lm[j]);
    }
    const bool any = ballot64(tie) != 0;
    if (lane_id() == 0) flags[k] = (uint8_t)((h.status != BBX_ST_OK ? 1 : 0) | (any ? 2 : 0));
  }
  if (lane_id() == 0) {
    if (!keep_counters) { h.need_reset = 0; h.budget = 0; h.rollout_pos = 0; h.t = 0; }
    if (seeds) { if (seed_std) h.std_rng = seeds[k]; else h.agent_seed = seeds[k]; }
    *(BbxHdr*)d = h;
  }
}
extern "C" int bbx_launch_clone(const char* src_recs, char* dst_recs, const BbxLayout* L, const int32_t* src, const int32_t* dst, int n,
                                const uint32_t* seeds, int keep_counters, int seed_std, int ngen, uint8_t* flags, hipStream_t stream) {
  const int blocks = (n + 3) / 4;
  if (L->W == 2) hipLaunchKernelGGL((bbx_clone_kernel<2>), dim3(blocks), dim3(256), 0, stream, src_recs, dst_recs, *L, src, dst, n, seeds, keep_counters, seed_std, ngen, flags);
  else if (L->W == 4) hipLaunchKernelGGL((bbx_clone_kernel<4>), dim3(blocks), dim3(256), 0, stream, src_recs, dst_recs, *L, src, dst, n, seeds, keep_counters, seed_std, ngen, flags);
  else hipLaunchKernelGGL((bbx_clone_kernel<8>), dim3(blocks), dim3(256), 0, stream, src_recs, dst_recs, *L, src, dst, n, seeds, keep_counters, seed_std, ngen, flags);
  return (int)hipGetLastError();
}
// ---- value(): the reducer order buchberger() starts from ---------------------------------------------------------------
// buchberger() re-sorts its reducers with std::sort (buchberger.cpp:157-158: G_ = F in basis order, sorted by lead
// monomial).  The environment keeps its own reducer order by upper_bound insertion, which is the STABLE order; the two
// differ only when lead monomials tie — possible among the ideal's generators only — AND the basis has more than 16
// elements, where libstdc++'s std::sort stops being an insertion sort: introsort (median-of-three quicksort with a depth
// limit of 2 log2 n and a heapsort fallback, segments of <= 16 left to a final insertion sort; bits/stl_algo.h
// __introsort_loop / __final_insertion_sort, GCC 11: ss_std_sort in bbx_device.h).  For the clones that need it one lane
// reproduces that sort on an index array, the wave then rebuilds the reducer-order arrays.
template <int W>
__global__ void bbx_value_resort_kernel(char* recs, BbxLayout L, int n, const uint8_t* flags) {
  const int k = blockIdx.x * (blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
  if (k >= n || !(flags[k] & 2)) return;
  char* rec = recs + (size_t)k * L.rec_bytes;
  const BbxHdr* h = (const BbxHdr*)rec;
  const int nG = h->nG, lane = lane_id();
  if (nG <= 16) return;                                    // std::sort is its (stable) insertion sort there: the environment's order
  const Mono<W>* lm = (const Mono<W>*)(rec + L.off_lm);
  uint16_t* ord = (uint16_t*)(rec + L.off_lcm);            // (the update's scratch: idle between steps)
  if (lane == 0) {
    for (int i = 0; i < nG; i++) ord[i] = (uint16_t)i;
    SortCtx<W> c{lm, ord};
    ss_std_sort<W>(c, nG);
  }
  wave_sync();
  Mono<W>* slm = (Mono<W>*)(rec + L.off_slm);
  if (L.kind == 1) {
    const Mono<W>* tm = (const Mono<W>*)(rec + L.off_tm);
    Mono<W>* stm = (Mono<W>*)(rec + L.off_stm);
    const uint2* gi = (const uint2*)(rec + L.off_ginfo);
    uint2* si = (uint2*)(rec + L.off_sinfo);
    for (int r = lane; r < nG; r += WAVE) {
      const int g = ord[r];
      const uint2 gg = gi[g];
      const uint32_t tc = gg.x >> 16, inv = gg.y & 0xffffu, sug = gg.y >> 16;
      slm[r] = lm[g]; stm[r] = tm[g];
      si[r] = make_uint2(tc | (negmod(mulmod(tc, inv)) << 16), sug | ((uint32_t)g << 16));   // .x = tc | (-tc / lc) << 16 (bin_add_poly)
    }
  } else {
    uint16_t* sidx = (uint16_t*)(rec + L.off_sidx);
    for (int r = lane; r < nG; r += WAVE) { const int g = ord[r]; slm[r] = lm[g]; sidx[r] = (uint16_t)g; }
  }
}
extern "C" int bbx_launch_value_resort(char* recs, const BbxLayout* L, int n, const uint8_t* flags, hipStream_t stream) {
  const int blocks = (n + 3) / 4;
  if (L->W == 2) hipLaunchKernelGGL((bbx_value_resort_kernel<2>), dim3(blocks), dim3(256), 0, stream, recs, *L, n, flags);
  else if (L->W == 4) hipLaunchKernelGGL((bbx_value_resort_kernel<4>), dim3(blocks), dim3(256), 0, stream, recs, *L, n, flags);
  else hipLaunchKernelGGL((bbx_value_resort_kernel<8>), dim3(blocks), dim3(256), 0, stream, recs, *L, n, flags);
  return (int)hipGetLastError();
}

// value(): per clone {discounted return, 0 when the rollout ran to the end without an error} after the rollouts
__global__ void bbx_value_collect_kernel(const char* recs, uint32_t rec_bytes, int n, double* out2) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const BbxHdr* h = (const BbxHdr*)(recs + (size_t)k * rec_bytes);
  out2[2 * (size_t)k] = h->vret;
  out2[2 * (size_t)k + 1] = (h->status != BBX_ST_OK || h->nP != 0) ? (double)(h->status ? h->status : -1) : 0.0;
}
extern "C" int bbx_launch_value_collect(const char* recs, uint32_t rec_bytes, int n, double* out2, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_value_collect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, recs, rec_bytes, n, out2);
  return (int)hipGetLastError();
}

// enlarged records (bbx_api.cpp grow_records): every environment's live state moves from its record in the old layout to
// its record in the new one; an environment that was waiting for room (bbx_st_capacity) is released
template <int W>
__global__ void bbx_relayout_kernel(const char* src_recs, char* dst_recs, BbxLayout Ls, BbxLayout Ld, int B) {
  const int k = blockIdx.x * (blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
  if (k >= B) return;
  char* s = const_cast<char*>(src_recs) + (size_t)k * Ls.rec_bytes;
  char* d = dst_recs + (size_t)k * Ld.rec_bytes;
  BbxHdr h = *(const BbxHdr*)s;
  if (Ls.kind == 1) bstage_copy<W>(benv_view<W>(d, Ld), benv_view<W>(s, Ls), h.nG, h.nP);
  else stage_copy<W>(env_view<W>(d, Ld), env_view<W>(s, Ls), h.nG, h.nP, h.arena_used);
  if (lane_id() == 0) {
    if (bbx_st_capacity(h.status)) h.status = BBX_ST_OK;
    *(BbxHdr*)d = h;
  }
}
extern "C" int bbx_launch_relayout(const char* src_recs, char* dst_recs, const BbxLayout* Ls, const BbxLayout* Ld, int B, hipStream_t stream) {
  const int blocks = (B + 3) / 4;
  if (Ls->W == 2) hipLaunchKernelGGL((bbx_relayout_kernel<2>), dim3(blocks), dim3(256), 0, stream, src_recs, dst_recs, *Ls, *Ld, B);
  else if (Ls->W == 4) hipLaunchKernelGGL((bbx_relayout_kernel<4>), dim3(blocks), dim3(256), 0, stream, src_recs, dst_recs, *Ls, *Ld, B);
  else hipLaunchKernelGGL((bbx_relayout_kernel<8>), dim3(blocks), dim3(256), 0, stream, src_recs, dst_recs, *Ls, *Ld, B);
  return (int)hipGetLastError();
}

// persistent sessions: the host's hand on the control word (one relaxed agent-scope atomic store: what the waves' polls observe)
__global__ void bbx_ctl_kernel(unsigned long long* ctl, unsigned long long value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) __hip_atomic_store(ctl, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
extern "C" int bbx_launch_ctl(unsigned long long* ctl, unsigned long long value, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_ctl_kernel, dim3(1), dim3(64), 0, stream, ctl, value);
  return (int)hipGetLastError();
}

// compact copy of every header so the host reads them with one contiguous transfer
__global__ void bbx_gather_hdr_kernel(const char* recs, uint32_t rec_bytes, int B, BbxHdr* out) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  out[env] = *(const BbxHdr*)(recs + (size_t)env * rec_bytes);
}
// the four header words the host polls after every launch
__global__ void bbx_gather_lite_kernel(const char* recs, uint32_t rec_bytes, int B, int4* out) {
  int env = blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= B) return;
  const BbxHdr* h = (const BbxHdr*)(recs + (size_t)env * rec_bytes);
  out[env] = make_int4(h->status, h->q_head, h->budget, h->nP);
}
extern "C" int bbx_launch_gather_lite(const char* recs, uint32_t rec_bytes, int B, void* out, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_gather_lite_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, recs, rec_bytes, B, (int4*)out);
  return (int)hipGetLastError();
}
extern "C" int bbx_launch_gather_hdr(const char* recs, uint32_t rec_bytes, int B, BbxHdr* out, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_gather_hdr_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, recs, rec_bytes, B, out);
  return (int)hipGetLastError();
}

// prepared weights of the PMLP policy kernels (layout: bbx_pmlp.h)
__global__ void bbx_pmlp_prepare_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2, float b2,
                                        int cols, int hidden, float* __restrict__ out) {
  const int HP = 32 * pmlp_nb_for(hidden), K2 = 2 * pmlp_ks_for(cols);
  const int total = (K2 + 2) * HP + 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int k = i / HP, h = i - k * HP;
    float v = 0.f;
    if (k < K2) v = (k < cols && h < hidden) ? w1[(size_t)k * hidden + h] : 0.f;
    else if (k == K2) v = h < hidden ? b1[h] : 0.f;
    else if (k == K2 + 1) v = h < hidden ? w2[h] : 0.f;
    else v = h == 0 ? b2 : 0.f;
    out[i] = v;
  }
}
extern "C" int bbx_launch_pmlp_prepare(const float* w1, const float* b1, const float* w2, float b2, int cols, int hidden, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_pmlp_prepare_kernel, dim3(16), dim3(256), 0, stream, w1, b1, w2, b2, cols, hidden, out);
  return (int)hipGetLastError();
}
extern "C" int bbx_launch_pmlp_act(const int32_t* obs, const int32_t* rows, int B, int obs_rows, int cols, const float* wp, int hidden, const float* u,
                                   int32_t* actions, float* logprobs, hipStream_t stream) {
  const int waves = 4, nb = pmlp_nb_for(hidden), ks = pmlp_ks_for(cols);
  const size_t ml = pmlp_lds_bytes(waves, obs_rows);
#define BBX_PMLP_MFMA(N, K) hipLaunchKernelGGL((bbx_pmlp_act_mfma_kernel<N, K>), dim3((B + waves - 1) / waves), dim3(waves * WAVE), ml, stream, obs, rows, B, \
                                                obs_rows, cols, wp, u, actions, logprobs)
#define BBX_PMLP_MFMA_K(N) do { if (ks == 3) BBX_PMLP_MFMA(N, 3); else if (ks == 6) BBX_PMLP_MFMA(N, 6); else if (ks == 10) BBX_PMLP_MFMA(N, 10); \
                                else if (ks == 16) BBX_PMLP_MFMA(N, 16); else BBX_PMLP_MFMA(N, 32); } while (0)
  if (nb == 1) BBX_PMLP_MFMA_K(1); else if (nb == 2) BBX_PMLP_MFMA_K(2); else if (nb == 4) BBX_PMLP_MFMA_K(4); else BBX_PMLP_MFMA_K(8);
#undef BBX_PMLP_MFMA_K
#undef BBX_PMLP_MFMA
  return (int)hipGetLastError();
}


// ------------------------------------------------------------------ host-callable launcher
// kind: 0 = HBM-resident step kernel, 1 = LDS-staged step kernel, 2 = aux (reset / observation only), 3 = the hand-tuned
// register/LDS-resident kernel (bbx_fast.h), 4 = wide (envs_per_block is then the number of waves per environment)
extern "C" int bbx_launch_step(const BbxParams* p, int kind, int envs_per_block, hipStream_t stream) {
  const int threads = envs_per_block * WAVE;
  const int blocks = (p->B + envs_per_block - 1) / envs_per_block;
  if (kind == 3) { bbx_launch_fast(p, blocks, threads, envs_per_block, stream); return (int)hipGetLastError(); }
  if (kind == 4) {
    if (p->L.W != 2 && p->L.W != 4 && p->L.W != 8) return (int)hipErrorInvalidValue;
    return bbx_launch_wide(p, envs_per_block, stream);
  }
  const size_t lds = kind == 1 ? (size_t)envs_per_block * p->LL.rec_bytes : 0;
  const int rc = p->L.kind == 1 ? bbx_launch_binom(p, kind, blocks, threads, lds, stream) : bbx_launch_general(p, kind, blocks, threads, lds, stream);
  if (rc) return rc;
  return (int)hipGetLastError();
}
