// Binomial class of the step kernel (kernels; bbx_binom.hip holds the launcher).
//
// For the random binomial distributions (3-20-10-weighted, 5-10-5-uniform, ... ; reference
// ideals.cpp:156-201) every basis element has at most two terms for the whole computation: an
// S-polynomial of binomials is a binomial, and reducing a binomial by binomials yields a binomial,
// a monomial or zero.  The class exploits that: no term arena, the S-polynomial h and the remainder
// r live in registers (two terms each), one reduction round is the ballot scan of the reducers' lead
// monomials plus two independent loads of the chosen reducer, and all merging is a handful of
// wave-uniform compares.  The observable results are identical to the general path (same tests).
//
// Reference semantics reproduced: buchberger.cpp:18-21 (spoly), 24-49 (reduce), 52-99 (update),
// 299-329 (reset/step), 354-408 (observation); polynomials.cpp:148-202 (merge, negate, term product).
#pragma once

#ifdef BBX_PROF_BUILD
static __device__ unsigned long long bbx_bin_prof_acc[32];
#define BSTAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); bprof[slot] += t_ - blast; blast = t_; } while (0)
#else
#define BSTAMP(slot) do {} while (0)
#endif

template <int W> struct BEnv {
  static constexpr bool kCached = false;
  BbxHdr* hdr;
  Mono<W>*lm, *tm, *slm, *stm, *lcm;
  uint2 *ginfo, *sinfo;
  uint32_t* pairs;
  uint8_t* cp;
};
template <int W> __device__ __forceinline__ BEnv<W> benv_view(char* rec, const BbxLayout& L) {
  BEnv<W> e;
  e.hdr = (BbxHdr*)rec;
  e.lm = (Mono<W>*)(rec + L.off_lm); e.tm = (Mono<W>*)(rec + L.off_tm); e.slm = (Mono<W>*)(rec + L.off_slm);
  e.stm = (Mono<W>*)(rec + L.off_stm); e.lcm = (Mono<W>*)(rec + L.off_lcm);
  e.ginfo = (uint2*)(rec + L.off_ginfo); e.sinfo = (uint2*)(rec + L.off_sinfo);
  e.pairs = (uint32_t*)(rec + L.off_pairs); e.cp = (uint8_t*)(rec + L.off_cp);
  return e;
}

// a term with c == 0 is "absent"
template <int W> struct BTerm { Mono<W> m; uint32_t c; };

// ---- HBM-resident environments with an LDS copy of what is gathered by index (16-byte monomials: 5-10-5-uniform) -------------
// The arrays the step reads by per-lane index — the pair list, and lm[] / tm[] behind the pair indices: pair filter of the
// update, Gebauer-Moeller lcms, observation rows — are what a wave waits for at 4 waves per SIMD: every such gather is a
// trip to L2 / HBM of a microsecond or two.  BEnvC keeps a WRITE-THROUGH copy of their first BC_G / BC_P entries in the wave's
// LDS: every write goes to the record as before (the record is complete at any moment: hand-over, growth and the end of the
// launch need nothing), reads of entries below the caps come from LDS.  Monomials are held one byte per slot (8 bytes
// instead of 16: two arrays of 480 fit beside 512 pairs in the 10 KB a wave can have at 16 environments per CU); a monomial
// with a slot beyond 255 switches the copy of lm / tm off until the next ideal (cap = 0: everything is read from the
// record again).  This replaces the peel scratch (wave_update forms the lcms from the copy where it reads them).
// Measured on one box (5-10-5-uniform, B = 4096 x 2048 steps): 106.6 M env-steps/s without the copy, 109-110 M with it; lm
// alone as plain 16-byte entries with the observation's gathers from the record: 100 M.
#define BBX_LDS __attribute__((address_space(3)))
constexpr int BC_G = 480, BC_P = 512;
constexpr int BC_BYTES = 2 * BC_G * 8 + BC_P * 4;          // 9728
typedef unsigned long long bc_u64;
__device__ __forceinline__ bc_u64 bc_pack(const Mono<4>& m) {  // (slot <= 255 each: the caller has checked)
  return (bc_u64)__builtin_amdgcn_perm(m.w[1], m.w[0], 0x06040200u) | ((bc_u64)__builtin_amdgcn_perm(m.w[3], m.w[2], 0x06040200u) << 32);
}
__device__ __forceinline__ Mono<4> bc_unpack(const bc_u64 p) {
  const uint32_t x = (uint32_t)p, y = (uint32_t)(p >> 32);
  Mono<4> m;
  m.w[0] = __builtin_amdgcn_perm(0u, x, 0x0c010c00u); m.w[1] = __builtin_amdgcn_perm(0u, x, 0x0c030c02u);
  m.w[2] = __builtin_amdgcn_perm(0u, y, 0x0c010c00u); m.w[3] = __builtin_amdgcn_perm(0u, y, 0x0c030c02u);
  return m;
}
__device__ __forceinline__ bool bc_fits(const Mono<4>& m) { return ((m.w[0] | m.w[1] | m.w[2] | m.w[3]) & 0xff00ff00u) == 0u; }
// N gathers at once — entry i[u] of array cs[u] / gs[u] (the copy and the record's array: the two views share the caps): all
// reads of the copy are in flight together, and so are the record's for the lanes whose index lies beyond the copy (one trip
// to memory for all of them instead of one per read, which is what N separate operator[] calls cost)
template <int N>
__device__ __forceinline__ void bc_gather(BBX_LDS bc_u64* const (&cs)[N], Mono<4>* const (&gs)[N], const int (&i)[N], int cap, Mono<4> (&o)[N]) {
  bc_u64 v[N]; bool far[N]; bool any = false;
#pragma unroll
  for (int u = 0; u < N; u++) { far[u] = i[u] >= cap; any |= far[u]; v[u] = cs[u][far[u] ? 0 : i[u]]; }
  if (ballot64(any)) {
#pragma unroll
    for (int u = 0; u < N; u++) { o[u] = m_zero<4>(); if (far[u]) o[u] = m_ld<4>(gs[u] + i[u]); }
  }
#pragma unroll
  for (int u = 0; u < N; u++) if (!far[u]) o[u] = bc_unpack(v[u]);
}
struct BCMono {                                            // read view of lm[] / tm[]
  Mono<4>* g; BBX_LDS bc_u64* c; int cap;
  __device__ __forceinline__ Mono<4> operator[](int i) const {
    if (i < cap) return bc_unpack(c[i]);
    return g[i];
  }
  // both members of four pair words (wave_update's filter): eight gathers in flight
  __device__ __forceinline__ void gather_pairs(const uint32_t (&pr)[4], Mono<4> (&li)[4], Mono<4> (&lj)[4]) const {
    BBX_LDS bc_u64* const cs[8] = {c, c, c, c, c, c, c, c}; Mono<4>* const gs[8] = {g, g, g, g, g, g, g, g};
    const int i[8] = {(int)(pr[0] & 0xffffu), (int)(pr[1] & 0xffffu), (int)(pr[2] & 0xffffu), (int)(pr[3] & 0xffffu),
                      (int)(pr[0] >> 16), (int)(pr[1] >> 16), (int)(pr[2] >> 16), (int)(pr[3] >> 16)};
    Mono<4> o[8];
    bc_gather<8>(cs, gs, i, cap, o);
#pragma unroll
    for (int u = 0; u < 4; u++) { li[u] = o[u]; lj[u] = o[4 + u]; }
  }
};
struct BCPairs {                                           // read / write view of the pair list
  uint32_t* g; BBX_LDS uint32_t* c;
  struct Ref {
    uint32_t* g; BBX_LDS uint32_t* c; int k;
    __device__ __forceinline__ operator uint32_t() const { return k < BC_P ? c[k] : g[k]; }
    __device__ __forceinline__ void operator=(uint32_t v) const { g[k] = v; if (k < BC_P) c[k] = v; }
  };
  __device__ __forceinline__ Ref operator[](int k) const { return Ref{g, c, k}; }
};
struct BEnvC {
  static constexpr bool kCached = true;
  BbxHdr* hdr;
  BCMono lm, tm;
  Mono<4>*slm, *stm, *lcm;
  uint2 *ginfo, *sinfo;
  BCPairs pairs;
  uint8_t* cp;
};
__device__ __forceinline__ BEnvC benvc_view(char* rec, const BbxLayout& L, char* lds) {
  BEnvC e;
  e.hdr = (BbxHdr*)rec;
  e.lm.g = (Mono<4>*)(rec + L.off_lm); e.tm.g = (Mono<4>*)(rec + L.off_tm);
  e.lm.c = (BBX_LDS bc_u64*)lds; e.tm.c = (BBX_LDS bc_u64*)(lds + BC_G * 8); e.lm.cap = e.tm.cap = 0;
  e.slm = (Mono<4>*)(rec + L.off_slm); e.stm = (Mono<4>*)(rec + L.off_stm); e.lcm = (Mono<4>*)(rec + L.off_lcm);
  e.ginfo = (uint2*)(rec + L.off_ginfo); e.sinfo = (uint2*)(rec + L.off_sinfo);
  e.pairs.g = (uint32_t*)(rec + L.off_pairs); e.pairs.c = (BBX_LDS uint32_t*)(lds + 2 * BC_G * 8);
  e.cp = (uint8_t*)(rec + L.off_cp);
  return e;
}
// fill the copy from the record (kernel start)
__device__ __forceinline__ void benv_load_cache(BEnvC& e, int nG, int nP) {
  const int lane = lane_id();
  bool fits = true;
  const int ng = nG < BC_G ? nG : BC_G;
  for (int i = lane; i < ng; i += WAVE) {
    const Mono<4> a = m_ld<4>(e.lm.g + i), b = m_ld<4>(e.tm.g + i);
    fits = fits && bc_fits(a) && bc_fits(b);
    e.lm.c[i] = bc_pack(a); e.tm.c[i] = bc_pack(b);
  }
  const int np = nP < BC_P ? nP : BC_P;
  for (int k = lane; k < np; k += WAVE) e.pairs.c[k] = e.pairs.g[k];
  e.lm.cap = e.tm.cap = ballot64(!fits) ? 0 : BC_G;
  wave_sync();
}
__device__ __forceinline__ void benv_load_cache(BEnv<2>&, int, int) {}
__device__ __forceinline__ void benv_load_cache(BEnv<4>&, int, int) {}
__device__ __forceinline__ void benv_load_cache(BEnv<8>&, int, int) {}
// a new ideal: the copy is empty and valid again
__device__ __forceinline__ void benv_new_ideal(BEnvC& e) { e.lm.cap = e.tm.cap = BC_G; }
template <int W> __device__ __forceinline__ void benv_new_ideal(BEnv<W>&) {}
// G[g] = (lead, tail): the basis-order arrays of the record, and the copy
__device__ __forceinline__ void benv_put(BEnvC& e, int g, const Mono<4>& lead, const Mono<4>& tail) {
  if (!(bc_fits(lead) && bc_fits(tail))) e.lm.cap = e.tm.cap = 0;            // (wave-uniform: the terms are)
  if (lane_id() == 0) {
    e.lm.g[g] = lead; e.tm.g[g] = tail;
    if (g < e.lm.cap) { e.lm.c[g] = bc_pack(lead); e.tm.c[g] = bc_pack(tail); }
  }
}
template <int W> __device__ __forceinline__ void benv_put(BEnv<W>& e, int g, const Mono<W>& lead, const Mono<W>& tail) {
  if (lane_id() == 0) { e.lm[g] = lead; e.tm[g] = tail; }
}

// (x) + (y) for single optional terms: polynomials.cpp:148-177 restricted to one term per side
template <int W>
__device__ __forceinline__ void merge2(const BTerm<W>& x, const BTerm<W>& y, BTerm<W>& o0, BTerm<W>& o1) {
  // select form (no branches): which of x / y leads, or both collapse into one term
  const bool hx = x.c != 0, hy = y.c != 0;
  const bool xgt = m_gt(x.m, y.m), ygt = m_gt(y.m, x.m);
  const bool same = hx && hy && !xgt && !ygt;             // equal monomials: sum, a zero sum drops the term
  const bool xfirst = hx && (!hy || xgt || same);
  const uint32_t sum = addmod(x.c, y.c);
  o0.c = same ? sum : (xfirst ? x.c : y.c);
  o1.c = (hx && hy && !same) ? (xfirst ? y.c : x.c) : 0u;
#pragma unroll
  for (int i = 0; i < W; i++) { o0.m.w[i] = xfirst ? x.m.w[i] : y.m.w[i]; o1.m.w[i] = xfirst ? y.m.w[i] : x.m.w[i]; }
}

template <int W>
__device__ void bstage_copy(const BEnv<W>& dst, const BEnv<W>& src, int nG, int nP) {
  const int lane = lane_id();
  for (int i = lane; i < nG; i += WAVE) {
    dst.lm[i] = src.lm[i]; dst.tm[i] = src.tm[i]; dst.slm[i] = src.slm[i]; dst.stm[i] = src.stm[i];
    dst.ginfo[i] = src.ginfo[i]; dst.sinfo[i] = src.sinfo[i];
  }
  for (int i = lane; i < nP; i += WAVE) dst.pairs[i] = src.pairs[i];
}

// append the binomial (t0, t1) as G[nG]: metadata, update(), sorted reducer insert (buchberger.cpp:321-326)
template <int W, class EnvB>
__device__ __forceinline__ bool bin_add_poly(EnvB& e, const BbxParams& p, const BbxLayout& L, int& nG, int& nP,
                             const BTerm<W>& t0, const BTerm<W>& t1, int sugar, int* status, char* peel_lds = nullptr,
                             unsigned long long* prof = nullptr, unsigned long long* plast = nullptr, int* first_drop = nullptr) {
  const int lane = lane_id();
  if (nG >= (int)L.maxG) { *status = BBX_ST_G_FULL; return false; }
  const int g = nG;
  const uint32_t inv = t0.c == 1 ? 1u : (uint32_t)uni((int)p.inv_table[t0.c]);   // 1/LC (polynomials.cpp:11-23)
  benv_put(e, g, t0.m, t1.m);
  if (lane == 0) e.ginfo[g] = make_uint2(t0.c | (t1.c << 16), inv | ((uint32_t)sugar << 16));
  wave_sync();
  if (!wave_update<W>(e, L, nG, nP, t0.m, p.elim, status, peel_lds, prof, plast, first_drop)) return false;
  // reducer order: std::upper_bound by lead monomial
  int pos = g;
  if (p.sort_reducers) {
    pos = 0;
    constexpr int UI = 4;
    for (int base = 0; base < g; base += WAVE * UI) {   // #reducers with LM <= LM f: four chunk loads in flight per trip
      Mono<W> sv[UI];
#pragma unroll
      for (int u = 0; u < UI; u++) { const int k = base + u * WAVE + lane; sv[u] = k < g ? e.slm[k] : m_zero<W>(); }
#pragma unroll
      for (int u = 0; u < UI; u++) { const int k = base + u * WAVE + lane; pos += __popcll(ballot64(k < g && !m_gt(sv[u], t0.m))); }
    }
    for (int hi = g; hi > pos; hi -= WAVE * UI) {      // shift [pos, g) up by one, from the top, four chunks per trip:
      Mono<W> a[UI], b[UI]; uint2 si[UI];               // all loads of a trip complete before its first store
#pragma unroll
      for (int u = 0; u < UI; u++) {
        const int k = hi - 1 - u * WAVE - lane;
        si[u] = make_uint2(0, 0); a[u] = m_zero<W>(); b[u] = m_zero<W>();
        if (k >= pos) { a[u] = m_ld<W>(e.slm + k); b[u] = m_ld<W>(e.stm + k); si[u] = e.sinfo[k]; }
      }
      wave_sync();
#pragma unroll
      for (int u = 0; u < UI; u++) {
        const int k = hi - 1 - u * WAVE - lane;
        if (k >= pos) { m_st<W>(e.slm + k + 1, a[u]); m_st<W>(e.stm + k + 1, b[u]); e.sinfo[k + 1] = si[u]; }
      }
      wave_sync();
    }
  }
  if (lane == 0) {
    e.slm[pos] = t0.m; e.stm[pos] = t1.m;
    e.sinfo[pos] = make_uint2(t1.c | (negmod(mulmod(t1.c, inv)) << 16), (uint32_t)sugar | ((uint32_t)g << 16));   // .x = tc | (-tc / lc) << 16
  }
  wave_sync();
  nG = g + 1;
  return true;
}

template <int W, class EnvB>
__device__ __forceinline__ bool bin_reset_inl(EnvB& e, const BbxParams& p, const BbxLayout& L, int env, int& nG, int& nP, int& q_head, int* status,
                                              uint32_t& gen_state, char* peel_lds = nullptr) {
  if (p.gen) {                                       // the ideal is drawn here (gen_binomial): no queue, no host
    const int npoly = (int)ldc(p.gen + 2), ncp = (int)ldc(p.gen + 4);
    const uint32_t gflags = ldc(p.gen + 3);
    const GenLanes GL = gen_lanes(p.gen);
    uint32_t x = (uint32_t)uni((int)gen_state);
    const int lane = lane_id();
    for (;;) {
      const uint32_t x_start = x;
      nG = 0; nP = 0;
      benv_new_ideal(e);
      // sort_input: all generators are drawn first (lane f keeps generator f), then enter in sorted order
      Mono<W> tabL = m_zero<W>(), tabT = m_zero<W>(); uint32_t tabC = 0; int rank = 0;
      if (p.sort_input) {
        for (int f = 0; f < npoly; f++) {
          Mono<W> lead, tail; uint32_t c;
          if (!gen_binomial<W>(x, p.gen, GL, gflags, ncp, lead, tail, c)) { *status = BBX_ST_GEN_FAIL; gen_state = x; return false; }
          if (lane == f) { tabL = lead; tabT = tail; tabC = c; }
        }
        rank = gen_sorted_rank<W>(tabL, npoly);
      }
      for (int f = 0; f < npoly; f++) {
        BTerm<W> t0, t1;
        t0.c = 1;
        if (p.sort_input) {
          const int src = __builtin_ctzll(ballot64(lane < npoly && rank == f));
#pragma unroll
          for (int q = 0; q < W; q++) { t0.m.w[q] = (uint32_t)__builtin_amdgcn_readlane((int)tabL.w[q], src); t1.m.w[q] = (uint32_t)__builtin_amdgcn_readlane((int)tabT.w[q], src); }
          t1.c = (uint32_t)__builtin_amdgcn_readlane((int)tabC, src);
        } else if (!gen_binomial<W>(x, p.gen, GL, gflags, ncp, t0.m, t1.m, t1.c)) { *status = BBX_ST_GEN_FAIL; gen_state = x; return false; }
        if (!bin_add_poly<W>(e, p, L, nG, nP, t0, t1, (int)m_deg(t0.m), status, peel_lds)) { gen_state = x_start; return false; }   // (a spill redoes this draw)
      }
      if (nP != 0) { gen_state = x; return true; }   // buchberger.cpp:313-314: redraw while the pair set is empty
    }
  }
  for (;;) {
    const uint32_t* slot;
    if (p.q.fixed) slot = p.q.words;
    else {
      const int tail = ldc(p.q.tail + env);
      if (q_head >= tail) { *status = BBX_ST_STARVED; return false; }
      slot = p.q.words + (size_t)env * p.q.env_stride + (size_t)(q_head % (int)p.q.nslots) * p.q.slot_words;
    }
    nG = 0; nP = 0;
    benv_new_ideal(e);
    const int npoly = (int)ldc(slot);               // queue words come in through scalar loads (see ldc)
    const uint32_t* w = slot + 1;
    for (int f = 0; f < npoly; f++) {
      const int n = (int)ldc(w), sugar = (int)ldc(w + 1);
      w += 2;
      if (n < 1 || n > 2) { *status = BBX_ST_POLY_TOO_LONG; return false; }
      BTerm<W> t0, t1;
      t0.c = ldc(w);
#pragma unroll
      for (int i = 0; i < W; i++) t0.m.w[i] = ldc(w + 1 + i);
      t1.c = 0; t1.m = m_zero<W>();
      if (n == 2) {
        t1.c = ldc(w + 1 + W);
#pragma unroll
        for (int i = 0; i < W; i++) t1.m.w[i] = ldc(w + 2 + W + i);
      }
      if (!bin_add_poly<W>(e, p, L, nG, nP, t0, t1, sugar, status, peel_lds)) return false;
      w += (size_t)n * (1 + W);
    }
    if (!p.q.fixed) q_head++;
    if (nP != 0 || p.q.fixed) return true;        // buchberger.cpp:313-314: redraw while the pair set is empty
  }
}

// out of line where nothing of the caller's has to live in memory for it (plain environments: read-only view); the cached view
// carries state (its caps) and is inlined, or the view, the kernel parameters and the counters would sit in scratch memory
template <int W, class EnvB>
__device__ bool bin_reset(EnvB& e, const BbxParams& p, const BbxLayout& L, int env, int& nG, int& nP, int& q_head, int* status,
                          uint32_t& gen_state, char* peel_lds = nullptr) {
  return bin_reset_inl<W>(e, p, L, env, nG, nP, q_head, status, gen_state, peel_lds);
}

// observation rows (buchberger.cpp:354-370, 391-394): one lane per monomial slot of the matrix
template <int W, bool HASH, class EnvB>
__device__ __forceinline__ uint64_t bin_obs(const EnvB& e, const BbxParams& p, int env, int nP, bool write, bool want_hash, char* stage_lds = nullptr,
                            size_t obs_off = 0, int row_from = 0) {
  // row_from > 0: rows [0, row_from) of the block already hold exactly these rows (the previous step of this launch wrote
  // them and the pair list has not changed in front of row_from): only the rows from there on are gathered and stored
  const int lane = lane_id();
  const int n = p.nvars, k = p.k;
  const int cols = 2 * n * k;
  int32_t* out = (write && p.obs) ? p.obs + obs_off + (size_t)env * p.obs_rows * cols : nullptr;
  const int rows = out ? (nP < p.obs_rows ? nP : p.obs_rows) : nP;
  // lane -> (row within the sweep, half, term) is fixed for the launch: no division inside the loop
  const int per_row = 2 * k;
  const int rows_per_sweep = per_row <= WAVE ? WAVE / per_row : 1;
  uint64_t h = 0;
  if (per_row <= WAVE) {
    const int rl = lane / per_row, slot = lane - rl * per_row;
    const int half = slot >= k ? 1 : 0, t = slot - half * k;
    const bool active = rl < rows_per_sweep;
    // four sweeps per trip with all gathers of a level in flight together: for HBM-resident environments each level is
    // a trip to L2/HBM, and a single dependent chain per trip left the wave idle for microseconds.  tm[g] is the
    // zero monomial when G[g] has no tail (bin_add_poly stores it that way), so no look at ginfo is needed.
    constexpr int U = 4;
    // (the pair words of the NEXT trip are requested before this trip's gathers: a trip then costs one dependent round trip
    // to memory — pair word -> monomial — instead of two: +4 %.  Measured and dropped this round: eight sweeps in flight through
    // direct-to-LDS gathers (global_load_lds_dwordx4: 81 M against 88 M), 16-byte aligned stores staged through LDS (52 M):
    // the observation is bound by the issue of its stores, not by the latency of its gathers.)
    uint32_t prn[U];
    const int r_begin = row_from - row_from % rows_per_sweep;         // (whole sweeps: a few unchanged rows are rewritten with what they hold)
#pragma unroll
    for (int u = 0; u < U; u++) { const int r = r_begin + u * rows_per_sweep + rl; prn[u] = (active && r < rows) ? e.pairs[r] : 0u; }
    for (int r0 = r_begin; r0 < rows; r0 += U * rows_per_sweep) {
      int rr[U]; bool on[U]; uint32_t pr[U]; Mono<W> mm[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        rr[u] = r0 + u * rows_per_sweep + rl;
        on[u] = active && rr[u] < rows;
        pr[u] = prn[u];
      }
#pragma unroll
      for (int u = 0; u < U; u++) { const int r = rr[u] + U * rows_per_sweep; prn[u] = (active && r < rows) ? e.pairs[r] : 0u; }
      if constexpr (EnvB::kCached) {
        // lane -> (lm or tm) is fixed: one gather per sweep from the lane's array (rows beyond |P| and terms beyond the second
        // read entry 0: never stored)
        BBX_LDS bc_u64* const cs_ = t == 0 ? e.lm.c : e.tm.c; Mono<4>* const gs_ = t == 0 ? e.lm.g : e.tm.g;
        BBX_LDS bc_u64* const cs[U] = {cs_, cs_, cs_, cs_}; Mono<4>* const gs[U] = {gs_, gs_, gs_, gs_};
        int gi_[U];
#pragma unroll
        for (int u = 0; u < U; u++) gi_[u] = (on[u] && t < 2) ? (half ? (int)(pr[u] >> 16) : (int)(pr[u] & 0xffffu)) : 0;
        bc_gather<U>(cs, gs, gi_, e.lm.cap, mm);
        if (t >= 2) {
#pragma unroll
          for (int u = 0; u < U; u++) mm[u] = m_zero<W>();
        }
      } else {
#pragma unroll
        for (int u = 0; u < U; u++) {
          const int g = half ? (int)(pr[u] >> 16) : (int)(pr[u] & 0xffffu);
          mm[u] = m_zero<W>();
          if (on[u]) { if (t == 0) mm[u] = e.lm[g]; else if (t == 1) mm[u] = e.tm[g]; }
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        if (on[u]) {
          const int base = (rr[u] * per_row + slot) * n;
          if (out) obs_store<W>(out + base, mm[u], n);
          if (HASH && want_hash) for (int v = 0; v < n; v++) h += bbx_mix64((uint64_t)(base + v), m_exp(mm[u], v));
        }
      }
    }
  } else {                                        // k > 32: generic item loop
    const int items = rows * per_row;
    for (int it = lane; it < items; it += WAVE) {
      int r = it / per_row, slot = it - r * per_row;
      int half = slot >= k ? 1 : 0, t = slot - half * k;
      uint32_t pr = e.pairs[r];
      int g = half ? (int)(pr >> 16) : (int)(pr & 0xffffu);
      Mono<W> mm = m_zero<W>();
      if (t == 0) mm = e.lm[g];
      else if (t == 1 && (e.ginfo[g].x >> 16) != 0) mm = e.tm[g];
      const int base = it * n;
      if (out) obs_store<W>(out + base, mm, n);
      if (HASH && want_hash) for (int v = 0; v < n; v++) h += bbx_mix64((uint64_t)(base + v), m_exp(mm, v));
    }
  }
  if (out && p.obs_fill) {
    for (int idx = rows * cols + lane; idx < p.obs_rows * cols; idx += WAVE) out[idx] = -1;
  }
  return (HASH && want_hash) ? wave_sum64(h) : 0;
}

// words of oracle/trace.py poly_words: [nterms, c0, e0[8], c1, e1[8]]
template <int W, class EnvB>
__device__ uint64_t bin_poly_hash(const EnvB& e, int g) {
  uint64_t h = 0;
  if (lane_id() == 0) {
    uint2 gi = e.ginfo[g];
    const uint32_t c0 = gi.x & 0xffffu, c1 = gi.x >> 16;
    const int n = c1 ? 2 : 1;
    h = bbx_mix64(0, (uint32_t)n);
    Mono<W> m0 = e.lm[g], m1 = e.tm[g];
    h += bbx_mix64(1, c0);
    for (int v = 0; v < BBX_MAXVARS; v++) h += bbx_mix64(2 + v, v < 2 * W - 1 ? m_exp(m0, v) : 0u);
    if (c1) {
      h += bbx_mix64(10, c1);
      for (int v = 0; v < BBX_MAXVARS; v++) h += bbx_mix64(11 + v, v < 2 * W - 1 ? m_exp(m1, v) : 0u);
    }
  }
  return wave_sum64(h);
}

// per-wave LDS scratch of the policy instantiations: the update's peel scratch, or the logits where those need more
template <int W> __host__ __device__ constexpr size_t binom_scratch_bytes(int obs_rows) {
  return (size_t)update_lds_bytes<W>() > sizeof(float) * (size_t)pmlp_lgcap(obs_rows) ? (size_t)update_lds_bytes<W>() : sizeof(float) * (size_t)pmlp_lgcap(obs_rows);
}
// POL > 0 (HBM-resident instantiation only): a policy rollout (bbx_policy_rollout_device) — the per-step protocol of
// fast_body POL (bbx_fast.h), rows gathered from the record — either as the continuation pass for environments that
// outgrew the register/LDS-resident class or as the rollout kernel of a batch that is not in that class.  POL = unit
// blocks of 32 of the hidden layer, PKS = the k-steps of the prepared weights (pmlp_ks_for(2 n k)).
// AUX: the instantiation without LDS (smem == nullptr).
template <int W, bool STAGED, bool TRACE, int POL = 0, int PKS = 6, bool AUX = false>
__device__ __forceinline__ void binom_body(const BbxParams& p_entry, char* smem, const BbxPolicy* pol = nullptr) {
  const BbxParams& p = bbx_kparams();                      // (set-up; the step loop and the write-back behind it take their own)
  (void)p_entry;
  const int lane = lane_id();
  const int wave_in_block = uni((int)(threadIdx.x / WAVE));
  const int env = blockIdx.x * (blockDim.x / WAVE) + wave_in_block;
  if (env >= p.B) return;
  char* grec = p.recs + (size_t)env * p.L.rec_bytes;
  BbxHdr* ghdr = (BbxHdr*)grec;
  const BbxLayout& L = STAGED ? p.LL : p.L;

  int nG = uni(ghdr->nG), nP = uni(ghdr->nP);
  int status = uni(ghdr->status), need_reset = uni(ghdr->need_reset), q_head = uni(ghdr->q_head);
  int t_agent = uni(ghdr->t), episode_steps = uni(ghdr->episode_steps);
  int episodes = uni(ghdr->episodes), zero_red = uni(ghdr->zero_reductions);
  long long total_steps = ghdr->total_steps, total_adds = ghdr->total_additions, alg_bytes = ghdr->alg_bytes;
  const uint32_t agent_seed = (uint32_t)uni((int)ghdr->agent_seed);
  uint32_t std_rng = (uint32_t)uni((int)ghdr->std_rng);
  uint32_t gen_state = ghdr->gen_rng;
  int budget = uni(ghdr->budget), rollout_pos = uni(ghdr->rollout_pos);
  int done_last = uni(ghdr->done_last);
  // behind a kernel of a persistent session this class only serves what that kernel handed over (an environment whose
  // slice was over waits for the session's next kernel)
  if (p.sess_target && status != BBX_ST_SPILL) return;
  const bool was_transient = status == BBX_ST_STARVED || status == BBX_ST_SPILL || status == BBX_ST_TIMESLICE;
  if (was_transient) status = BBX_ST_OK;
  double vret = ghdr->vret, vdisc = ghdr->vdisc;
  int obs_trunc = uni(ghdr->obs_trunc);
  if (p.set_budget) { budget = bbx_st_capacity(status) ? budget + p.nsteps : p.nsteps; rollout_pos = 0; done_last = 0; vret = 0.0; vdisc = 1.0; obs_trunc = 0; }   // (bbx_common.h: bbx_st_capacity)
  if (p.sess_target) budget = p.sess_target - uni(ghdr->sess_done);   // closing launch of a persistent session: what is still owed
  if (p.pass == 1 && !(status == BBX_ST_OK && (need_reset || (budget > 0 && nP > 0)))) {
    // nothing to do here.  If the kernel in front could not hold the environment (it is too large for the register/LDS
    // class) although there was nothing to take either, its hand-over mark must not outlive the launch: the host would
    // keep resuming an environment that has no step left
    if (was_transient && status == BBX_ST_OK && lane == 0) {
      ghdr->status = BBX_ST_OK;
      if (p.lite) p.lite[4 * (size_t)env] &= ~0xffff;
    }
    return;
  }

  // the HBM-resident instantiation for 16-byte monomials works through an LDS copy of what it gathers by index (BEnvC); the
  // launcher provides BC_BYTES per wave for it, the Gebauer-Moeller peel scratch (update_lds_bytes) for the others
  constexpr bool CACHE = W == 4 && !STAGED && POL == 0 && !AUX;
  typedef typename std::conditional<CACHE, BEnvC, BEnv<W>>::type EnvB;
  BEnv<W> ge = benv_view<W>(grec, p.L);
  EnvB e;
  if constexpr (CACHE) e = benvc_view(grec, p.L, smem + (size_t)wave_in_block * BC_BYTES);
  else e = STAGED ? benv_view<W>(smem + (size_t)wave_in_block * L.rec_bytes, L) : ge;
  if (CACHE && status == BBX_ST_OK) benv_load_cache(e, nG, nP);
  bool staged_in = false;
  if (STAGED && status == BBX_ST_OK && !(!need_reset && nP == 0)) {   // (an idle environment is not staged: bbx_fast.h, idle0)
    if (nG > (int)L.maxG || nP > (int)L.maxP) status = BBX_ST_SPILL;
    else { if constexpr (STAGED) bstage_copy<W>(e, ge, nG, nP); staged_in = true; wave_sync(); }
  }
  int steps_done = 0;
  bool obs_live = false;                                   // the caller's block holds the observation of the state as of the last step
  double last_reward = 0.0;
  const bool tracing = TRACE && p.trace != nullptr;
  const int obs_term_bytes = 4 * 2 * p.nvars * p.k;
  // HBM-resident instantiation: the launcher provides one Gebauer-Moeller peel scratch per wave in LDS
  // (policy kernels: the scratch also holds the logits of up to pmlp_lgcap(obs_rows) rows — binom_scratch_bytes)
  char* const peel_lds = (!STAGED && !CACHE && smem != nullptr) ? smem + (size_t)wave_in_block * (POL > 0 ? binom_scratch_bytes<W>(p.obs_rows) : (size_t)update_lds_bytes<W>()) : nullptr;
#ifdef BBX_PROF_BUILD
  unsigned long long bprof[32] = {0};
  unsigned long long blast = __builtin_amdgcn_s_memtime();
#endif

  for (;;) {
    const BbxParams& p = bbx_kparams();                    // (per step: nothing of it lives across the loop's back edge)
    const BbxLayout& L = STAGED ? p.LL : p.L;
    pin(nG, nP, status, need_reset, q_head, t_agent, episode_steps, episodes, zero_red, budget, rollout_pos, done_last, obs_trunc, steps_done);
    pin(total_steps, total_adds, alg_bytes, std_rng, gen_state, vret, vdisc, last_reward, obs_live);   // (bbx_device.h: pin)
    if (status != BBX_ST_OK) break;
    if (need_reset) {
      bool reset_ok;
      if constexpr (CACHE) reset_ok = bin_reset_inl<W>(e, p, L, env, nG, nP, q_head, &status, gen_state, peel_lds);
      else reset_ok = bin_reset<W>(e, p, L, env, nG, nP, q_head, &status, gen_state, peel_lds);
      if (!reset_ok) {
        if (STAGED && (status == BBX_ST_G_FULL || status == BBX_ST_P_FULL)) { status = BBX_ST_SPILL; nG = 0; nP = 0; }
        break;
      }
      need_reset = 0; episode_steps = 0; obs_live = false;
      if constexpr (POL > 0) {
        // (per-step policy calls served by a session: the block the caller finds afterwards is the NEW episode's — bbx_fast.h)
        if (pol->post_obs) {
          if (p.obs) { bin_obs<W, false>(e, p, env, nP, true, false, nullptr); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
          if (lane == 0 && pol->rows_t) pol->rows_t[env] = nP;
        }
      }
    }
    if (budget <= 0) break;
    if (nP == 0) break;
    if (nG + 1 > (int)L.maxG || nP - 1 + nG > (int)L.maxP) {          // worst-case headroom, before any mutation
      status = STAGED ? BBX_ST_SPILL : (nG + 1 > (int)L.maxG ? BBX_ST_G_FULL : BBX_ST_P_FULL);
      break;
    }

    BSTAMP(0);
    // ---- choose the pair ------------------------------------------------------------------------
    int action;
    int pol_tt = 0;
    if constexpr (POL > 0) {
      pol_tt = uni((p.sess_target ? p.sess_target : p.nsteps) - budget);   // (a continuation: the budget carries on from the first pass;
                                                                            // behind a session's kernel: what is owed of the session's total)
      const size_t tb = (size_t)pol_tt * (size_t)p.B + (size_t)env;
      const float uu = pol->u[tb];
      const bool pre_obs = pol->post_obs == 0;
      if (p.obs && pre_obs) { bin_obs<W, false>(e, p, env, nP, true, false, nullptr, (size_t)pol_tt * (size_t)pol->obs_tstride); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
      if (lane == 0 && pol->rows_t && pre_obs) pol->rows_t[tb] = nP;
      int n = nP < PMLP_MAXROWS ? nP : PMLP_MAXROWS;            // (rows beyond what the policy can score: reported, like rows
      obs_trunc |= nP > PMLP_MAXROWS ? 1 : 0;                    // beyond the caller's block — bbx_sync returns BBX_E_CAPACITY)
      if (p.obs) n = n < p.obs_rows ? n : p.obs_rows;
      const float* wp = pol->wp;
      const int plr = lane & 31, plk = lane >> 5;
      float* lg = (float*)peel_lds;                          // (the update's scratch is idle here: PMLP_MAXROWS floats)
      const float b2 = wp[(size_t)(2 * PKS + 2) * 32 * POL];
      // column c = 2 s + (lane >> 5) of a row = exponent (c mod n) of term (c / n) of [lead_i, tail_i, .., lead_j, tail_j, ..]
      // (k terms per polynomial, zeros beyond a binomial's two): decoded once, the same for every row
      const int pn_ = p.nvars, pk_ = p.k;
      int csel[PKS];
#pragma unroll
      for (int s2 = 0; s2 < PKS; s2++) {
        const int c = 2 * s2 + plk;
        const int t = c / pn_, v = c - t * pn_, member = t / pk_, which = t - member * pk_;
        csel[s2] = (c < 2 * pn_ * pk_ && which < 2) ? (v | (which << 8) | (member << 9)) : -1;
      }
      for (int r0 = 0; r0 < n; r0 += 32) {
        const int r = r0 + plr;
        const uint32_t prw = r < n ? e.pairs[r] : 0u;
        const Mono<W> a0 = e.lm[prw & 0xffffu], a1 = e.tm[prw & 0xffffu], c0 = e.lm[prw >> 16], c1 = e.tm[prw >> 16];
        float xa[PKS];
#pragma unroll
        for (int s2 = 0; s2 < PKS; s2++) {
          const int cs = csel[s2];
          Mono<W> mm;
#pragma unroll
          for (int q = 0; q < W; q++) { const uint32_t wi = (cs & 256) ? a1.w[q] : a0.w[q], wj = (cs & 256) ? c1.w[q] : c0.w[q]; mm.w[q] = (cs & 512) ? wj : wi; }
          xa[s2] = cs < 0 ? 0.f : (float)m_exp(mm, cs & 255);
        }
        const float logit = pmlp_tile<POL, PKS, 1>(xa, wp, plr, plk);
        if (plk == 0 && r < n) lg[r] = logit + b2;
      }
      wave_sync();
      action = pmlp_sample(lg, n, env, uu, pol->actions + (size_t)pol_tt * (size_t)pol->stride_out, pol->logprobs + (size_t)pol_tt * (size_t)pol->stride_out);
    } else
    if (p.agent == BBX_AGENT_EXTERNAL) action = p.actions[env];
    else if (p.agent == BBX_AGENT_HASH) action = (int)bbx_agent_action32(agent_seed, (uint32_t)t_agent, (uint32_t)nP);
    else if (p.agent == BBX_AGENT_FIRST) action = 0;
    else if (p.agent == BBX_AGENT_LAST) action = nP - 1;
    else if (p.agent == BBX_AGENT_STDRANDOM) action = std_choice(std_rng, nP);
    else if constexpr (CACHE) action = select_pair_inl<W>(e, nP, p.agent, [&](int g) { return (int)(e.ginfo[g].y >> 16); });
    else action = select_pair<W>(e, nP, p.agent, [&](int g) { return (int)(e.ginfo[g].y >> 16); });
    action = uni(action);
    if (action < 0 || action >= nP) { status = BBX_ST_BAD_ACTION; break; }
    const uint32_t pr = (uint32_t)uni((int)e.pairs[action]);
    const int gi = pr & 0xffffu, gj = pr >> 16;
    // The reducers' lead monomials do not change during a reduction: the first 64 * SR of them (reducer order) are
    // loaded ONCE, all loads in flight together, and every round scans registers; larger bases continue in memory.
    // (Requested here, ahead of the pair removal and the S-polynomial, whose own trips to memory they then overlap.)
    constexpr int SR = W == 2 ? 8 : (W == 4 ? 6 : 3);
    Mono<W> S[SR];
    const int nsr = (nG + WAVE - 1) / WAVE < SR ? (nG + WAVE - 1) / WAVE : SR;
#pragma unroll
    for (int u = 0; u < SR; u++) { const int k = u * WAVE + lane; S[u] = (u < nsr && k < nG) ? m_ld<W>(e.slm + k) : m_zero<W>(); }
    for (int base = action; base < nP - 1; base += WAVE * 4) {         // P.erase(remove(action)), stable: four chunks per trip
      uint32_t v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = base + u * WAVE + lane; v[u] = k < nP - 1 ? e.pairs[k + 1] : 0u; }
      wave_sync();
#pragma unroll
      for (int u = 0; u < 4; u++) { const int k = base + u * WAVE + lane; if (k < nP - 1) e.pairs[k] = v[u]; }
      wave_sync();
    }
    nP -= 1;

    BSTAMP(1);
    // ---- S-polynomial (buchberger.cpp:18-21): the lead terms cancel, the two tails remain -------------
    BTerm<W> h0, h1;
    int hsug;
    int bytes = 0;                                                     // algorithmic bytes of this step
    {
      const Mono<W> lmi = e.lm[gi], lmj = e.lm[gj];
      const uint2 ii = e.ginfo[gi], ij = e.ginfo[gj];
      const Mono<W> gamma = m_lcm(lmi, lmj);
      const Mono<W> si = m_div(gamma, lmi), sj = m_div(gamma, lmj);
      BTerm<W> a, b;
      const uint32_t tci = ii.x >> 16, tcj = ij.x >> 16;
      a.c = tci ? mulmod(tci, ii.y & 0xffffu) : 0u;               a.m = m_mul(e.tm[gi], si);
      b.c = tcj ? negmod(mulmod(tcj, ij.y & 0xffffu)) : 0u;       b.m = m_mul(e.tm[gj], sj);
      int sgi = (int)(ii.y >> 16) + (int)m_deg(si), sgj = (int)(ij.y >> 16) + (int)m_deg(sj);
      hsug = uni(sgi > sgj ? sgi : sgj);
      if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; break; }
      merge2<W>(a, b, h0, h1);
      bytes += 12 * ((tci ? 2 : 1) + (tcj ? 2 : 1) + (h0.c ? 1 : 0) + (h1.c ? 1 : 0));
    }

    BSTAMP(2);
    // ---- reduce (buchberger.cpp:24-49) --------------------------------------------------------------
    BTerm<W> r0, r1;
    r0.c = 0; r1.c = 0; r0.m = m_zero<W>(); r1.m = m_zero<W>();
    int nsteps_red = 0, rsug = 0;
    bool overflow = false;
    while (uni((int)h0.c) != 0) {
      const int hn = h1.c ? 2 : 1;
      int found = -1;
      Mono<W> lmg = m_zero<W>();
#pragma unroll
      for (int u = 0; u < SR; u++) {                                    // first reducer whose LM divides LM(h)
        if (u < nsr && found < 0) {
          const uint64_t mask = ballot64(u * WAVE + lane < nG && m_divides(S[u], h0.m));
          if (mask) {
            const int src = __builtin_ctzll(mask);
            found = u * WAVE + src;
#pragma unroll
            for (int i = 0; i < W; i++) lmg.w[i] = (uint32_t)__builtin_amdgcn_readlane((int)S[u].w[i], src);
          }
        }
      }
      for (int base = SR * WAVE; base < nG && found < 0; base += WAVE) {
        int k = base + lane;
        Mono<W> s = m_zero<W>();
        bool d = false;
        if (k < nG) { s = e.slm[k]; d = m_divides(s, h0.m); }
        uint64_t mask = ballot64(d);
        if (mask) {
          const int src = __builtin_ctzll(mask);
          found = base + src;
#pragma unroll
          for (int i = 0; i < W; i++) lmg.w[i] = (uint32_t)__builtin_amdgcn_readlane((int)s.w[i], src);
        }
      }
      if (found >= 0) {                                                 // h <- h - (LT h / LT f) f
        const uint2 si = e.sinfo[found];
        const Mono<W> tmg = e.stm[found];
        const uint32_t tcg = si.x & 0xffffu, kg = si.x >> 16;       // kg = -tc / lc mod p (0 without a tail)
        const Mono<W> q = m_div(h0.m, lmg);
        BTerm<W> b;
        b.c = mulmod(h0.c, kg);
        b.m = m_mul(tmg, q);
        int fs = (int)(si.y & 0xffffu) + (int)m_deg(q);
        hsug = uni(fs > hsug ? fs : hsug);
        if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; overflow = true; break; }
        BTerm<W> n0, n1;
        merge2<W>(h1, b, n0, n1);
        bytes += 8 * (found + 1) + 12 * (tcg ? 2 : 1) + 12 * (hn + (n0.c ? 1 : 0) + (n1.c ? 1 : 0));
        h0 = n0; h1 = n1;
        nsteps_red++;
        if (nsteps_red > (1 << 24)) { status = BBX_ST_RUNAWAY; overflow = true; break; }
      } else {                                                          // r <- r + LT h ; h <- h - LT h
        bytes += 8 * nG + 12 * (2 * hn - 1);
        if (r0.c == 0) r0 = h0; else r1 = h0;
        int d = (int)m_deg(h0.m);
        rsug = d > rsug ? d : rsug;
        h0 = h1; h1.c = 0;
      }
    }
    if (overflow) break;
    rsug = uni(rsug > hsug ? rsug : hsug);

    BSTAMP(3);
    // ---- basis / pair-set update (buchberger.cpp:321-327) ---------------------------------------------
    const int nG_before = nG, nP_before = nP;
    int first_drop = nP;
    if (uni((int)r0.c) != 0) {
#ifdef BBX_PROF_BUILD
      if (!bin_add_poly<W>(e, p, L, nG, nP, r0, r1, rsug, &status, peel_lds, bprof, &blast, &first_drop)) break;
      BSTAMP(4);
#else
      if (!bin_add_poly<W>(e, p, L, nG, nP, r0, r1, rsug, &status, peel_lds, nullptr, nullptr, &first_drop)) break;
#endif
      bytes += 12 * (r1.c ? 2 : 1) + 8 * nG_before + 8 * (nP_before + nP);
    } else zero_red++;
    bytes += nP * obs_term_bytes;
    alg_bytes += bytes;
    const double reward = (p.rewards_mode == BBX_REW_ADDITIONS) ? (-1.0 - (double)nsteps_red) : -1.0;
    last_reward = reward;
    if (p.value_mode) value_accumulate(vret, vdisc, reward, p.gamma);
    total_steps++; total_adds += 1 + nsteps_red; t_agent++; episode_steps++; steps_done++;
    const bool done = nP == 0;

    BSTAMP(5);
    if (POL == 0 && p.obs_every_step && p.obs) {
      // The block holds the previous step's observation (obs_live: this launch wrote it, no new ideal since): the pair list is
      // unchanged in front of the selected pair and of the first pair the update dropped, so those rows stand as they are —
      // with the random agent about half of them.  The matrix in memory after the step is the full observation either way.
      const int from = (obs_live && !p.obs_fill) ? uni(action < first_drop ? action : first_drop) : 0;
      bin_obs<W, false>(e, p, env, nP, true, false, peel_lds, 0, from);
      obs_trunc |= nP > p.obs_rows ? 1 : 0;
      obs_live = true;
    }
    if constexpr (POL > 0) {
      if (pol->post_obs && p.obs) { bin_obs<W, false>(e, p, env, nP, true, false, nullptr); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
      if (lane == 0) {
        const size_t tb = (size_t)pol_tt * (size_t)pol->stride_out + (size_t)env;
        if (pol->rewards_t) pol->rewards_t[tb] = reward;
        if (pol->dones_t) pol->dones_t[tb] = done ? 1 : 0;
        if (pol->post_obs && pol->rows_t) pol->rows_t[env] = nP;
      }
    }
    BSTAMP(6);
    if (TRACE && tracing) {
      uint64_t oh = bin_obs<W, true>(e, p, env, nP, false, true);
      uint64_t ph = wave_pairs_hash<W, EnvB>(e, nP);
      uint64_t nh = nG > nG_before ? bin_poly_hash<W>(e, nG - 1) : 0;
      if (lane == 0) {
        BbxTraceRec& tr = p.trace[(size_t)env * p.trace_stride + rollout_pos];
        tr.action = action; tr.nP = nP; tr.nG = nG; tr.done = done ? 1 : 0; tr.reward = reward;
        tr.obs_hash = oh; tr.pairs_hash = ph; tr.newpoly_hash = nh;
      }
    }
    budget--; rollout_pos++;
    done_last = done ? 1 : 0;
    if (done) {
      episodes++;
      if (p.auto_reset) need_reset = 1;
      // behind a kernel of a persistent session: the episode that outgrew the register/LDS class is over — the next one
      // starts small, so the environment goes back to the session's next kernel with what it still owes
      // (only where the host looks after it: slice_ticks != 0)
      if (p.sess_target && p.slice_ticks && p.auto_reset && budget > 0) { status = BBX_ST_TIMESLICE; break; }
    }
  }

#ifdef BBX_PROF_BUILD
  if (lane == 0) for (int i = 0; i < 32; i++) if (bprof[i]) atomicAdd(&bbx_bin_prof_acc[i], bprof[i]);
#endif
  const BbxParams& pz = bbx_kparams();                    // (the write-back)
  const bool handoff = status == BBX_ST_SPILL;
  if (POL == 0 && pz.obs && status == BBX_ST_OK) { bin_obs<W, false>(e, pz, env, nP, true, false, peel_lds); obs_trunc |= nP > pz.obs_rows ? 1 : 0; }
  if (STAGED && staged_in) {
    wave_sync();
    if constexpr (STAGED) bstage_copy<W>(ge, e, nG, nP);
  }
  if (lane == 0) {
    BbxHdr* h = ghdr;
    h->nG = nG; h->nP = nP; h->arena_used = 0; h->status = status; h->need_reset = need_reset;
    h->q_head = q_head; h->t = t_agent; h->std_rng = std_rng; h->episode_steps = episode_steps; h->total_steps = total_steps;
    h->gen_rng = gen_state;
    h->total_additions = total_adds; h->episodes = episodes; h->zero_reductions = zero_red; h->steps_done = steps_done;
    h->budget = budget; h->rollout_pos = rollout_pos; h->done_last = done_last; h->alg_bytes = alg_bytes;
    h->vret = vret; h->vdisc = vdisc; h->obs_trunc = obs_trunc;
    if (pz.sess_target) h->sess_done = pz.sess_target - budget;
    if (pz.lite) *(int4*)(pz.lite + 4 * (size_t)env) = make_int4(status | (obs_trunc ? BBX_LITE_OBS_TRUNC : 0), q_head, budget, nP);
    if (pz.value_mode && pz.values) pz.values[env] = vret;
    if (!handoff) {
      if (pz.rewards && (steps_done > 0 || pz.pass == 0)) pz.rewards[env] = last_reward;
      if (pz.dones) pz.dones[env] = (uint8_t)((done_last || (nP == 0 && !need_reset)) ? 1 : 0);
      if (pz.rows) pz.rows[env] = nP;
    }
  }
}

template <int W, bool STAGED, bool TRACE>
__global__ __launch_bounds__(256, 4) void bbx_binom_kernel(BbxParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  binom_body<W, STAGED, TRACE>(p, smem);
}
template <int W, int NB, int KS>
__global__ __launch_bounds__(256, 4) void bbx_binom_policy_kernel(BbxParams p, BbxPolicy pol) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  binom_body<W, false, false, NB, KS>(p, smem, &pol);
}
template <int W>
__global__ __launch_bounds__(256) void bbx_binom_aux_kernel(BbxParams p) {
  binom_body<W, false, false, 0, 6, true>(p, nullptr);
}
