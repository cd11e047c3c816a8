// Wide class of the step kernel (kernels; bbx_wide.hip holds the launcher): ONE WORKGROUP PER ENVIRONMENT, for environments whose
// polynomials are long and whose number is small (fixed ideals such as cyclic-n; BASELINE config "cyclic-7, batch 512").
//
// Reference semantics reproduced bit for bit: buchberger.cpp:18-21 (spoly), 24-49 (reduce: first divisor in G_ order,
// steps counted), 52-99 (update), 299-329 (reset / step), 354-408 (observation); polynomials.cpp:148-202 (operator+,
// operator-, Term * Polynomial).
//
// All waves of the workgroup run the step loop in lockstep; every decision about the environment (|G|, |P|, the
// polynomial being reduced, the divisor found) is computed from the same data by every wave, so control flow is
// workgroup-uniform and phases are separated by s_barrier only.  Where the state lives during a step:
//
//   LDS   the polynomial h being reduced, ping/pong (H0, H1: HC terms each) and the window F (FC terms) through which
//         the tail of the current reducer — already multiplied by the quotient term — streams.  A term in LDS is an
//         8-byte SORT KEY (grevlex a > b <=> key(a) > key(b) as unsigned integers: degree byte on top, exponent bytes
//         complemented below; 8-byte monomials of <= 3 variables as they are, 16-byte ones of <= 7 variables packed to
//         bytes while the sugar degree stays <= 255) plus a u16 coefficient: 10 bytes, so 80 KB hold h up to 2816 terms;
//         the reducers' lead monomials and per-reducer metadata IN REDUCER ORDER (first RC reducers), staged once per
//         basis change (north_star: "LDS-staged reducer lead monomials")
//   HBM   the basis (term arena), pair set and everything persistent (same record layout as the general class); the
//         remainder r is appended term by term directly behind the arena's end, so a non-zero reduction adds its
//         result to the basis without a copy
//
// One reduction round h <- h - (LT h / LT f) f:
//   1. first-divisor scan: thread k tests R[k] | LM(h) (ds_read_b128 + v_pk_sub_u16 clamp), ballot per wave, minimum
//      over waves through LDS                                                                       (1 barrier)
//   2. the reducer's tail is read from HBM/L2 (coalesced, its address comes from the LDS table: ONE dependent trip to
//      memory per round), multiplied by the quotient term and stored to F as keys                   (1 barrier)
//   3. merge-path: every thread binary-searches its diagonal of (h tail, F) in LDS, merges its <= SEG + 1 terms
//      sequentially (equal monomials summed mod p, zero sums dropped), positions by ballot prefix + a per-wave count
//      through LDS, result written to the other H buffer                                            (2 barriers / tile)
// Tiers, chosen per round by size: (1) all of the above in LDS; (2) h longer than HC lives in the record's HBM scratch
// and streams through LDS window by window against the fully staged reducer tail, the result streaming back out;
// (3) anything else (exponents beyond a byte, a reducer tail longer than the LDS) runs the same merge code on
// HBM-resident views.  Capacity is therefore a performance cliff, never a failure.
#pragma once

#define BBX_AS3 __attribute__((address_space(3)))
typedef uint32_t bbx_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t bbx_u32x4 __attribute__((ext_vector_type(4)));

constexpr int WSEG = 4;                                   // merged positions per thread per tile
constexpr int WSEG_SHORT = 2;                             // ... of short merges (see wide_merge)
constexpr int WNWMAX = 8;                                 // waves per workgroup (512 threads)
constexpr int WRB = 256;                                  // terms of the remainder r buffered in LDS between flushes to the arena

// Per-environment bookkeeping that nothing inside a step branches on lives in LDS (one writer: thread 0), not in
// registers: the step loop is long and every value live across it costs a register in all eight waves.
struct WideCold {
  long long total_steps, total_adds, alg_bytes;
  double vret, vdisc, last_reward;
  int episode_steps, episodes, zero_red, steps_done, rollout_pos, done_last, obs_trunc, q_head;
  uint32_t std_rng, gen_state;
  uint32_t rng_mark;                                      // std_rng before the step in progress (restored when the step has to be taken again)
};
struct __attribute__((aligned(16))) WideCtl {             // LDS control block (parity double-buffered exchange slots)
  int wfound[2][WNWMAX];                                  // per-wave first divisor of a scan chunk (absent waves: INT_MAX)
  int wcount[2][WNWMAX];                                  // per-wave output count of a merge tile (absent waves: 0)
  int wcnt4[2][4][WNWMAX];                                // per-wave output counts of the four sub-tiles of a per-element merge tile
  int wend[2][WNWMAX][2];                                 // per-wave last merge-path boundary of a tile
  int wany[2][WNWMAX];                                    // per-wave mask of scan candidates with a divisor (absent waves: 0)
  int bc[16];                                             // values the leader wave publishes to the workgroup
  WideCold st;
};
typedef int bbx_i32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int wide_ctl_bytes() { return (int)((sizeof(WideCtl) + 255) / 256 * 256); }
// LDS bytes of one workgroup: control block, 2*hc + fc terms (8-byte key + u16 coefficient), rc reducer table entries
__host__ __device__ constexpr size_t wide_lds_bytes(int W, int hc, int fc, int rc, int sc) {
  return (size_t)wide_ctl_bytes() + (size_t)(2 * hc + fc + WRB + 2 * sc) * 8 + (((size_t)(2 * hc + fc + WRB + 2 * sc) * 2 + 15) & ~(size_t)15) +
         (size_t)rc * (4 * W + 12);
}

// ---- sort keys --------------------------------------------------------------------------------------------------------
// grevlex a > b (polynomials.cpp:60-74)  <=>  wide_key(a) > wide_key(b): total degree in the top byte / halfword, below
// it the exponents from the LAST variable down, complemented (the smaller exponent wins).
__device__ __forceinline__ uint64_t wide_key(const Mono<2>& m) { return (((uint64_t)m.w[1] << 32) | m.w[0]) ^ 0x0000FFFFFFFFFFFFull; }
__device__ __forceinline__ Mono<2> wide_unkey(uint64_t k, Mono<2>*) {
  k ^= 0x0000FFFFFFFFFFFFull;
  Mono<2> m; m.w[0] = (uint32_t)k; m.w[1] = (uint32_t)(k >> 32);
  return m;
}
// 16-byte monomials (7 exponents + degree as u16) whose entries all fit a byte: the caller guarantees degree <= 255
__device__ __forceinline__ uint64_t wide_key(const Mono<4>& m) {
  const uint32_t lo = (m.w[0] & 0xffu) | ((m.w[0] >> 8) & 0xff00u) | ((m.w[1] & 0xffu) << 16) | ((m.w[1] << 8) & 0xff000000u);
  const uint32_t hi = (m.w[2] & 0xffu) | ((m.w[2] >> 8) & 0xff00u) | ((m.w[3] & 0xffu) << 16) | ((m.w[3] << 8) & 0xff000000u);
  return (((uint64_t)hi << 32) | lo) ^ 0x00FFFFFFFFFFFFFFull;
}
__device__ __forceinline__ Mono<4> wide_unkey(uint64_t k, Mono<4>*) {
  k ^= 0x00FFFFFFFFFFFFFFull;
  const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
  Mono<4> m;
  m.w[0] = (lo & 0xffu) | ((lo & 0xff00u) << 8); m.w[1] = ((lo >> 16) & 0xffu) | ((lo >> 8) & 0xff0000u);
  m.w[2] = (hi & 0xffu) | ((hi & 0xff00u) << 8); m.w[3] = ((hi >> 16) & 0xffu) | ((hi >> 8) & 0xff0000u);
  return m;
}
// 32-byte monomials (the reference's N = 8, polynomials.h:29) have no 8-byte key: for them the kernel works in its unkeyed
// regime throughout — the one 16-byte monomials enter when an exponent passes a byte: h as plain monomials in the record's
// scratch, lead terms fetched one by one, merges on HBM-resident views (tier 3).  The two functions below exist so that the
// keyed code compiles; wide_keys_ok() keeps every path that would call them closed.
__device__ __forceinline__ uint64_t wide_key(const Mono<8>&) { return 0; }
__device__ __forceinline__ Mono<8> wide_unkey(uint64_t, Mono<8>*) { return m_zero<8>(); }
template <int W> __device__ __forceinline__ Mono<W> wide_unkey(uint64_t k) { return wide_unkey(k, (Mono<W>*)nullptr); }
// can terms of sugar degree <= hsug be held as keys?  (8-byte monomials are their own key; 16-byte ones pack while every
// slot fits a byte, and the sugar degree bounds them all)
template <int W> __device__ __forceinline__ bool wide_keys_ok(int hsug) { return W == 2 || (W == 4 && hsug <= 255); }
__device__ __forceinline__ bool wk_gt(uint64_t a, uint64_t b) { return a > b; }
__device__ __forceinline__ bool wk_eq(uint64_t a, uint64_t b) { return a == b; }
template <int W> __device__ __forceinline__ bool wk_gt(const Mono<W>& a, const Mono<W>& b) { return m_gt(a, b); }
template <int W> __device__ __forceinline__ bool wk_eq(const Mono<W>& a, const Mono<W>& b) { return m_eq(a, b); }
__device__ __forceinline__ void wk_zero(uint64_t& k) { k = 0; }
template <int W> __device__ __forceinline__ void wk_zero(Mono<W>& k) { k = m_zero<W>(); }

// ---- accessors: the merge core is written once against these -------------------------------------------------------
struct LdsKeys {                                          // a polynomial in LDS: sort keys and u16 coefficients
  typedef uint64_t K;
  BBX_AS3 uint64_t* k; BBX_AS3 uint16_t* c;
  __device__ __forceinline__ uint64_t key(int i) const { return k[i]; }
  __device__ __forceinline__ uint32_t coef(int i) const { return c[i]; }
  __device__ __forceinline__ void put(int i, uint64_t kk, uint32_t cc) const { k[i] = kk; c[i] = (uint16_t)cc; }
  __device__ __forceinline__ LdsKeys off(int n) const { LdsKeys r; r.k = k + n; r.c = c + n; return r; }
};
template <int W> struct GlbView {                         // a polynomial in HBM seen through a term multiplier
  typedef Mono<W> K;
  const Mono<W>* m; const uint16_t* c; Mono<W> shift; uint32_t scale;
  __device__ __forceinline__ Mono<W> key(int i) const { return m_mul(m[i], shift); }
  __device__ __forceinline__ uint32_t coef(int i) const { return mulmod(c[i], scale); }
};
template <int W> struct GlbOut {                          // a polynomial being written to HBM
  Mono<W>* m; uint16_t* c;
  __device__ __forceinline__ void put(int i, const Mono<W>& mm, uint32_t cc) const { m[i] = mm; c[i] = (uint16_t)cc; }
  __device__ __forceinline__ void put(int i, uint64_t kk, uint32_t cc) const { m[i] = wide_unkey<W>(kk); c[i] = (uint16_t)cc; }
};

// everything the cooperative routines need to know about the workgroup
struct WideCtx {
  BBX_AS3 WideCtl* ctl;
  int tid, lane, wave, NT, NW;
  int par_found, par_count, par_end;                      // parities of the exchange slots (advance once per use)
#ifdef BBX_PROF_BUILD
  int prof_trips;
  unsigned long long mt[5], ml;                           // merge phases: search, boundary exchange, sequential part, count exchange, stores
#endif
};
#ifdef BBX_PROF_BUILD
#define MSTAMP0() do { x.ml = __builtin_amdgcn_s_memtime(); } while (0)
#define MSTAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); x.mt[k] += t_ - x.ml; x.ml = t_; } while (0)
#else
#define MSTAMP0() do {} while (0)
#define MSTAMP(k) do {} while (0)
#endif

// a value every lane holds alike (an LDS or memory load from a uniform address), pinned to scalar registers: the compiler
// takes every load for per-lane, and with it each comparison, branch and counter that depends on one
__device__ __forceinline__ uint64_t uni64(uint64_t v) {
  return ((uint64_t)(uint32_t)uni((int)(uint32_t)(v >> 32)) << 32) | (uint32_t)uni((int)(uint32_t)v);
}
__device__ __forceinline__ int wide_shr1(int v) {         // lane i <- lane i-1 (lane 0: undefined, fixed by the caller)
  return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, false);
}

// number of leading elements of the descending key sequence A[0..n) that are >= x (workgroup-uniform binary search)
__device__ __forceinline__ int wide_count_ge(const LdsKeys& A, int n, uint64_t x) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (A.key(mid) >= x) lo = mid + 1; else hi = mid;
  }
  return uni(lo);
}

// O[nout ...) <- A[0..na) + B[0..nb)  (both descending grevlex; equal monomials summed, zero sums dropped):
// Polynomial operator+ (polynomials.cpp:148-177) as a workgroup merge-path merge.  Returns the new output count; nothing
// is stored at or beyond ocap (the caller compares the count with ocap).  Called by ALL threads with uniform arguments.
template <int SEG, class AV, class BV, class OV>
__device__ __forceinline__ int wide_merge_seg(const AV& A, int na_, const BV& B, int nb_, const OV& O, int nout_, int ocap_, WideCtx& x) {
  typedef typename AV::K K;
  // uniform by construction; pinned so that the loops and branches on them are scalar
  const int na = uni(na_), nb = uni(nb_), ocap = uni(ocap_);
  int nout = uni(nout_);
  const int total = na + nb;
  int ci = 0, cj = 0;                                     // merge-path boundary at the start of the current tile
  for (int base = 0; base < total; base += x.NT * SEG) {
    x.par_end = uni(x.par_end); x.par_count = uni(x.par_count);
    MSTAMP0();
    // ---- my END boundary: (i1, j1), i1 + j1 = d, such that A[0..i1) and B[0..j1) are exactly the first d terms of the
    // merge (ties: the A term first); an equal pair is never split across a boundary
    int d = base + (x.tid + 1) * SEG; d = d < total ? d : total;
    int lo = d - nb > 0 ? d - nb : 0, hi = d < na ? d : na;
    // (binary search on purpose: four- and nine-way searches — 3 or 8 probes in flight per trip to LDS — were measured
    // 7 % and 16 % slower on one environment and 25-47 % slower at 512: the probes' instructions and registers cost more
    // than the trips they save)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (!wk_gt(B.key(d - 1 - mid), A.key(mid))) lo = mid + 1; else hi = mid;    // A[mid] >= B[d-1-mid]: beyond mid
    }
    int i1 = lo, j1 = d - lo;
    if (i1 > 0 && j1 < nb && wk_eq(A.key(i1 - 1), B.key(j1))) j1++;
    MSTAMP(0);
    // ---- my START boundary = the END boundary of the thread before me
    const int pe = x.par_end; x.par_end ^= 1;
    if (x.lane == 63) { x.ctl->wend[pe][x.wave][0] = i1; x.ctl->wend[pe][x.wave][1] = j1; }
    int i0 = wide_shr1(i1), j0 = wide_shr1(j1);
    __syncthreads();
    if (x.lane == 0) {
      if (x.wave == 0) { i0 = ci; j0 = cj; }
      else { i0 = x.ctl->wend[pe][x.wave - 1][0]; j0 = x.ctl->wend[pe][x.wave - 1][1]; }
    }
    ci = uni(x.ctl->wend[pe][x.NW - 1][0]); cj = uni(x.ctl->wend[pe][x.NW - 1][1]);
    MSTAMP(1);
    // ---- sequential merge of A[i0..i1) with B[j0..j1): at most SEG + 1 elements, at most SEG + 1 outputs
    K om[SEG + 1]; uint32_t oc[SEG + 1];
    int i = i0, j = j0;
    K a, b; wk_zero(a); wk_zero(b);
    uint32_t ac = 0, bc = 0;
    if (i < i1) { a = A.key(i); ac = A.coef(i); }
    if (j < j1) { b = B.key(j); bc = B.coef(j); }
#pragma unroll
    for (int s = 0; s <= SEG; s++) {
      const bool ha = i < i1, hb = j < j1;
      const bool bgt = wk_gt(b, a);
      const bool eq = ha && hb && wk_eq(a, b);
      const bool takeA = ha && (!hb || !bgt);             // A >= B, or B exhausted
      const bool takeB = hb && !takeA;
      uint32_t c = takeA ? ac : bc;
      if (eq) c = addmod(ac, bc);
      om[s] = takeA ? a : b;
      oc[s] = (ha || hb) ? c : 0u;                        // 0 marks "nothing here" (exhausted, or the sum cancelled)
      if (takeA) { i++; if (i < i1) { a = A.key(i); ac = A.coef(i); } }
      if (takeB || eq) { j++; if (j < j1) { b = B.key(j); bc = B.coef(j); } }
    }
    MSTAMP(2);
    // ---- positions: thread-major, within a thread in merge order
    int prefix = 0, wtot = 0;
#pragma unroll
    for (int s = 0; s <= SEG; s++) {
      const uint64_t mk = ballot64(oc[s] != 0);
      prefix += (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
      wtot += __popcll(mk);
    }
    const int pc = x.par_count; x.par_count ^= 1;
    if (x.lane == 0) x.ctl->wcount[pc][x.wave] = wtot;
    __syncthreads();
    int woff = 0, ttot = 0;
    {                                                       // all eight counts with two 16-byte reads (slots of absent waves hold 0)
      const bbx_i32x4 c0 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wcount[pc][0], c1 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wcount[pc][4];
      const int cw[WNWMAX] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
      for (int w = 0; w < WNWMAX; w++) { woff += w < x.wave ? cw[w] : 0; ttot += cw[w]; }
    }
    int pos = nout + woff + prefix;
    MSTAMP(3);
#pragma unroll
    for (int s = 0; s <= SEG; s++) {
      if (oc[s] != 0) { if (pos < ocap) O.put(pos, om[s], oc[s]); pos++; }
    }
    nout = uni(nout + ttot);
    MSTAMP(4);
  }
  return nout;
}

// The same merge for two polynomials held as keys in LDS, one OUTPUT position per thread and sub-tile: thread t of sub-tile k
// searches the diagonal d = base + k NT + t of the merge path — i = the number of A terms among the first d of the merge,
// ties to A — and the term at position d is then A[i] or B[d - i] by one comparison: A[i] with the coefficients summed when
// B[d - i] is the same monomial, nothing when it is a B term whose monomial A[i - 1] already carried.  No boundary travels
// between neighbouring threads and no thread merges sequentially: per tile one exchange (the waves' counts) instead of two,
// and the K searches of a thread run interleaved (K loads in flight per trip to LDS).
template <int K, class OV>
__device__ __forceinline__ int wide_merge_el(const LdsKeys& A, int na_, const LdsKeys& B, int nb_, const OV& O, int nout_, int ocap_, WideCtx& x) {
  const int na = uni(na_), nb = uni(nb_), ocap = uni(ocap_);
  int nout = uni(nout_);
  const int total = na + nb;
  for (int base = 0; base < total; base += x.NT * K) {
    x.par_count = uni(x.par_count);
    MSTAMP0();
    int lo[K], hi[K], dd[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int d = base + k * x.NT + x.tid;
      dd[k] = d < total ? d : total;
      lo[k] = dd[k] - nb > 0 ? dd[k] - nb : 0; hi[k] = dd[k] < na ? dd[k] : na;
    }
    for (;;) {
      bool any = false;
      uint64_t ak[K], bk[K];
#pragma unroll
      for (int k = 0; k < K; k++) {
        const bool go = lo[k] < hi[k];
        any = any || go;
        const int mid = (lo[k] + hi[k]) >> 1;
        ak[k] = A.key(go ? mid : 0); bk[k] = B.key(go ? dd[k] - 1 - mid : 0);
      }
      if (!any) break;
#pragma unroll
      for (int k = 0; k < K; k++) {
        const bool go = lo[k] < hi[k];
        const int mid = (lo[k] + hi[k]) >> 1;
        if (go) { if (ak[k] >= bk[k]) lo[k] = mid + 1; else hi[k] = mid; }       // A[mid] is among the first d terms of the merge
      }
    }
    MSTAMP(0);
    uint64_t om[K]; uint32_t oc[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const int i = lo[k], j = dd[k] - lo[k];
      const bool ha = i < na, hb = j < nb;
      const uint64_t a = A.key(ha ? i : 0), b = B.key(hb ? j : 0), ap = A.key(i > 0 ? i - 1 : 0);
      const uint32_t ca = A.coef(ha ? i : 0), cb = B.coef(hb ? j : 0);
      const bool takeA = ha && (!hb || a >= b);
      uint32_t c;
      if (takeA) c = (hb && a == b) ? addmod(ca, cb) : ca;
      else c = (hb && !(i > 0 && ap == b)) ? cb : 0u;
      om[k] = takeA ? a : b;
      oc[k] = base + k * x.NT + x.tid < total ? c : 0u;               // 0: nothing here (beyond the end, merged into A, or cancelled)
    }
    MSTAMP(2);
    int prefix[K], wtot[K];
#pragma unroll
    for (int k = 0; k < K; k++) {
      const uint64_t mk = ballot64(oc[k] != 0);
      prefix[k] = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u));
      wtot[k] = __popcll(mk);
    }
    const int pc = x.par_count; x.par_count ^= 1;
    if (x.lane == 0) {
#pragma unroll
      for (int k = 0; k < K; k++) x.ctl->wcnt4[pc][k][x.wave] = wtot[k];
    }
    __syncthreads();
    int pos0 = nout;
#pragma unroll
    for (int k = 0; k < K; k++) {
      const bbx_i32x4 c0 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wcnt4[pc][k][0], c1 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wcnt4[pc][k][4];
      const int cw[WNWMAX] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
      int woff = 0, ttot = 0;
#pragma unroll
      for (int w = 0; w < WNWMAX; w++) { woff += w < x.wave ? cw[w] : 0; ttot += cw[w]; }
      const int pos = pos0 + woff + prefix[k];
      if (oc[k] != 0 && pos < ocap) O.put(pos, om[k], oc[k]);
      pos0 = uni(pos0 + ttot);
    }
    MSTAMP(3);
    nout = pos0;
    MSTAMP(4);
  }
  return nout;
}

// SEG merged positions per thread per tile: short merges (the common case when one environment owns the whole device:
// a few hundred terms) are spread over more threads, which shortens each thread's sequential part
#ifndef BBX_WIDE_SEGMERGE
template <class OV>
__device__ __forceinline__ int wide_merge(const LdsKeys& A, int na, const LdsKeys& B, int nb, const OV& O, int nout, int ocap, WideCtx& x) {
  // (one position per thread while the merge fits one tile that way: the Degree run of cyclic-7 5.77 -> 5.50 s; long merges
  // keep one search per SEG positions — a search per output position costs more than it saves there: with K = 2 / 4 sub-tiles
  // cyclic-7 under the random agent ran at 37.7 k instead of 42.6 k env-steps/s, the Degree run at 5.82 s with K = 2)
  const int total = uni(na + nb);
  if (total <= x.NT) return wide_merge_el<1>(A, na, B, nb, O, nout, ocap, x);
  if (total <= x.NT * WSEG_SHORT) return wide_merge_seg<WSEG_SHORT>(A, na, B, nb, O, nout, ocap, x);
  return wide_merge_seg<WSEG>(A, na, B, nb, O, nout, ocap, x);
}
#endif
template <class AV, class BV, class OV>
__device__ __forceinline__ int wide_merge(const AV& A, int na, const BV& B, int nb, const OV& O, int nout, int ocap, WideCtx& x) {
  if (uni(na + nb) <= x.NT * WSEG_SHORT) return wide_merge_seg<WSEG_SHORT>(A, na, B, nb, O, nout, ocap, x);
  return wide_merge_seg<WSEG>(A, na, B, nb, O, nout, ocap, x);
}

// tier 3 (cold): the same merge on HBM-resident views.  Not inlined: its register needs (16-byte monomials in flight)
// must not weigh on the LDS tiers.  The context travels by value; the caller's exchange-slot parities stay valid
// because a barrier follows every call.
template <int W>
__device__ __noinline__ int wide_merge_hbm(const Mono<W>* am, const uint16_t* ac, int an, const Mono<W>* fm, const uint16_t* fc, int fn,
                                           Mono<W> shift, uint32_t scale, Mono<W>* om, uint16_t* oc, int ocap, WideCtx x) {
  GlbView<W> A, Bv;
  A.m = am; A.c = ac; A.shift = m_zero<W>(); A.scale = 1u;
  Bv.m = fm; Bv.c = fc; Bv.shift = shift; Bv.scale = scale;
  const GlbOut<W> O{om, oc};
  return wide_merge_seg<WSEG>(A, an, Bv, fn, O, 0, ocap, x);
}

// D[0..n) <- keys of (scale * x^shift) * src[0..n)   (Term * Polynomial, polynomials.cpp:196-202), all threads
template <int W>
__device__ __forceinline__ void wide_load_scaled(const LdsKeys& D, const Mono<W>* sm, const uint16_t* sc, int n,
                                                 const Mono<W>& shift, uint32_t scale, const WideCtx& x) {
  for (int t = x.tid; t < n; t += x.NT) {
    const Mono<W> mm = sm[t];
    const uint32_t cc = sc[t];
    D.put(t, wide_key(m_mul(mm, shift)), mulmod(cc, scale));
  }
}
// D[0..n) <- keys of src[0..n), all threads
template <int W>
__device__ __forceinline__ void wide_load_plain(const LdsKeys& D, const Mono<W>* sm, const uint16_t* sc, int n, const WideCtx& x) {
  for (int t = x.tid; t < n; t += x.NT) {
    const Mono<W> mm = sm[t];
    const uint32_t cc = sc[t];
    D.put(t, wide_key(mm), cc);
  }
}

// reducer table in LDS, reducer order: lead monomial (as stored: u16 slots, what the divisibility test wants) and
// {arena offset, #terms | sugar << 16, 1/LC | basis index << 16}
template <int W> struct WideTable {
  BBX_AS3 uint32_t* lm; BBX_AS3 bbx_u32x2* meta; BBX_AS3 uint32_t* poff;
  __device__ __forceinline__ Mono<W> mono(int i) const {
    Mono<W> r;
    if constexpr (W == 2) { const bbx_u32x2 v = *(BBX_AS3 bbx_u32x2*)(lm + 2 * i); r.w[0] = v.x; r.w[1] = v.y; }
    else {
#pragma unroll
      for (int q = 0; q < W / 4; q++) {
        const bbx_u32x4 v = *(BBX_AS3 bbx_u32x4*)(lm + W * i + 4 * q); r.w[4 * q] = v.x; r.w[4 * q + 1] = v.y; r.w[4 * q + 2] = v.z; r.w[4 * q + 3] = v.w;
      }
    }
    return r;
  }
  __device__ __forceinline__ void put_mono(int i, const Mono<W>& mm) const {
    if constexpr (W == 2) { bbx_u32x2 v; v.x = mm.w[0]; v.y = mm.w[1]; *(BBX_AS3 bbx_u32x2*)(lm + 2 * i) = v; }
    else {
#pragma unroll
      for (int q = 0; q < W / 4; q++) {
        bbx_u32x4 v; v.x = mm.w[4 * q]; v.y = mm.w[4 * q + 1]; v.z = mm.w[4 * q + 2]; v.w = mm.w[4 * q + 3]; *(BBX_AS3 bbx_u32x4*)(lm + W * i + 4 * q) = v;
      }
    }
  }
};

// first reducer (reducer order) whose lead monomial divides lmh, or -1: buchberger.cpp:29-33.  The first rcl reducers
// are tested in LDS, the rest (basis larger than the LDS table) in HBM/L2.
template <int W>
__device__ int wide_find_divisor(const WideTable<W>& R, int rcl_, const Mono<W>* slm, int nG_, const Mono<W>& lmh, WideCtx& x) {
  const int rcl = uni(rcl_), nG = uni(nG_);               // (uniform by construction; pinned: scalar loop)
  for (int base = 0; base < nG; base += x.NT) {
    const int k = base + x.tid;
    bool d = false;
    if (k < nG) {
      Mono<W> s;
      if (k < rcl) s = R.mono(k); else s = slm[k];
      d = m_divides(s, lmh);
    }
#ifdef BBX_PROF_BUILD
    x.prof_trips++;
#endif
    const uint64_t mask = ballot64(d);
    const int mine = mask ? base + x.wave * WAVE + (int)__builtin_ctzll(mask) : 0x7fffffff;
    const int pf = x.par_found; x.par_found ^= 1;
    if (x.lane == 0) x.ctl->wfound[pf][x.wave] = mine;
    __syncthreads();
    int found;
    {                                                       // (slots of absent waves hold INT_MAX)
      const bbx_i32x4 f0 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wfound[pf][0], f1 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wfound[pf][4];
      const int m0 = f0.x < f0.y ? f0.x : f0.y, m1 = f0.z < f0.w ? f0.z : f0.w, m2 = f1.x < f1.y ? f1.x : f1.y, m3 = f1.z < f1.w ? f1.z : f1.w;
      const int m4 = m0 < m1 ? m0 : m1, m5 = m2 < m3 ? m2 : m3;
      found = uni(m4 < m5 ? m4 : m5);
    }
    if (found != 0x7fffffff) return found;
  }
  return -1;
}

// Which of the nc <= 4 monomials c[] have a divisor among the reducers at all?  (bit k of the result.)  One pass over the
// table and ONE exchange for all of them: a run of irreducible terms — terms that go to the remainder one after the
// other, buchberger.cpp:41-44 — costs one scan per four terms instead of one each.
template <int W>
__device__ __forceinline__ int wide_reducible_mask(const WideTable<W>& R, int rcl_, const Mono<W>* slm, int nG_, const Mono<W> (&c)[4], int nc,
                                                   const WideCtx& x, int& par_any) {
  const int rcl = uni(rcl_), nG = uni(nG_);
  int mine = 0;
  for (int base = 0; base < nG; base += x.NT) {
    const int k = base + x.tid;
    if (k < nG) {
      Mono<W> s;
      if (k < rcl) s = R.mono(k); else s = slm[k];
#pragma unroll
      for (int j = 0; j < 4; j++) mine |= (j < nc && m_divides(s, c[j])) ? (1 << j) : 0;
    }
  }
  int wmask = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) wmask |= ballot64((mine >> j) & 1) ? (1 << j) : 0;
  const int pa = par_any; par_any ^= 1;
  if (x.lane == 0) x.ctl->wany[pa][x.wave] = wmask;
  __syncthreads();
  const bbx_i32x4 a0 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wany[pa][0], a1 = *(BBX_AS3 bbx_i32x4*)&x.ctl->wany[pa][4];
  return uni(a0.x | a0.y | a0.z | a0.w | a1.x | a1.y | a1.z | a1.w);
}

// lead-monomial observation (buchberger.cpp:354-370, 391-394), all threads: one item = one monomial slot of the matrix
// kernel arguments are re-read from the kernarg segment where they are used, through a pointer the optimiser cannot see
// through (constant address space + uniform address = s_load from the scalar cache): otherwise every field is loaded at
// kernel entry and stays live across the whole step loop
__device__ __forceinline__ const BbxParams& wide_params() {
  const __attribute__((address_space(4))) BbxParams* q = (const __attribute__((address_space(4))) BbxParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return *(const BbxParams*)q;                            // (the kernels take ONE by-value struct: offset 0)
}

template <int W>
__device__ void wide_obs(const Env<W>& e, const BbxParams& p, int env, int nP, const WideCtx& x) {
  const int n = p.nvars, k = p.k;
  const int cols = 2 * n * k;
  int32_t* out = p.obs + (size_t)env * p.obs_rows * cols;
  const int rows = nP < p.obs_rows ? nP : p.obs_rows;
  const int items = rows * 2 * k;
  for (int it = x.tid; it < items; it += x.NT) {
    const int r = it / (2 * k), rem = it - r * 2 * k;
    const int half = rem / k, t = rem - half * k;
    const uint32_t pr = e.pairs[r];
    const int g = half ? (int)(pr >> 16) : (int)(pr & 0xffffu);
    const bool have = t < (int)e.plen[g];
    const Mono<W> mm = have ? e.am[e.poff[g] + t] : m_zero<W>();
    obs_store<W>(out + it * n, mm, n);
  }
  if (p.obs_fill) for (int idx = rows * cols + x.tid; idx < p.obs_rows * cols; idx += x.NT) out[idx] = -1;
}

#ifdef BBX_PROF_BUILD   // diagnostic build only: cycles per phase and event counts, summed over all workgroups
__device__ unsigned long long bbx_wide_prof_acc[32];
// six coarse time slots and a few counters, all at compile-time indices (a 32-slot array lives in scratch and more than
// doubles the run time, which distorts the very shares it is meant to show)
#define WSLOT(slot) ((slot) == 2 ? 0 : (slot) == 4 ? 1 : (slot) == 20 ? 2 : (slot) == 22 ? 3 : ((slot) == 5 || (slot) == 6) ? 4 : 5)
#define WSTAMP(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); wprof[WSLOT(slot)] += t_ - wlast; wlast = t_; } while (0)
#define WCSLOT(slot) ((slot) == 10 ? 0 : (slot) == 13 ? 1 : (slot) == 23 ? 2 : (slot) == 24 ? 3 : (slot) == 25 ? 4 : (slot) == 15 ? 5 : 6)
#define WCOUNT(slot, v) (wcnt[WCSLOT(slot)] += (unsigned long long)(v))
#elif defined(BBX_MARK)
#define WSTAMP(slot) asm volatile("; WMARK " #slot)      // assembly listings only: phase boundaries for instruction counts
#define WCOUNT(slot, v) do {} while (0)
#else
#define WSTAMP(slot) do {} while (0)
#define WCOUNT(slot, v) do {} while (0)
#endif

// Leader-wave sections that run once per episode / once per step on the HBM record.  They are compiled as functions of
// their own (results travel through the control block) so that the register allocation of the reduction loop does not
// depend on code it never runs: an addition to the reset path once cost the loop 25 % through relocated spills.
template <int W>
__device__ __noinline__ void wide_leader_reset(const BbxParams* pp, BBX_AS3 WideCtl* ctl, int env) {
  const BbxParams& p = *pp;   // (the kernarg pointer intrinsic is null outside the kernel function itself)
  Env<W> e = env_view<W>(p.recs + (size_t)env * p.L.rec_bytes, p.L);
  BBX_AS3 WideCold* const st = &ctl->st;
  int nG = 0, nP = 0, arena_used = 0, status = BBX_ST_OK;
  int q_head = st->q_head; uint32_t gen_state = st->gen_state;
  const bool ok = wave_reset<W>(e, p, p.L, env, nG, nP, arena_used, q_head, &status, gen_state);
  if (lane_id() == 0) {
    ctl->bc[0] = ok ? 1 : 0; ctl->bc[1] = nG; ctl->bc[2] = nP; ctl->bc[3] = arena_used; ctl->bc[5] = status;
    st->q_head = q_head; st->gen_state = gen_state; st->episode_steps = 0;
  }
}
template <int W>
__device__ __noinline__ void wide_leader_add(const BbxParams* pp, BBX_AS3 WideCtl* ctl, int env, int nG, int nP, int arena_used, int rn, int rsug) {
  const BbxParams& p = *pp;   // (the kernarg pointer intrinsic is null outside the kernel function itself)
  Env<W> e = env_view<W>(p.recs + (size_t)env * p.L.rec_bytes, p.L);
  int status = BBX_ST_OK;
  wave_sync();
  const bool ok = wave_add_poly<W>(e, p.L, nG, nP, arena_used, e.am + arena_used, e.ac + arena_used, rn, rsug, p.elim, p.sort_reducers, &status, true);
  if (lane_id() == 0) { ctl->bc[0] = ok ? 1 : 0; ctl->bc[1] = nG; ctl->bc[2] = nP; ctl->bc[3] = arena_used; ctl->bc[5] = status; }
}
template <int W>
__device__ __noinline__ void wide_leader_select(const BbxParams* pp, BBX_AS3 WideCtl* ctl, int env, int nP, int agent) {
  const BbxParams& p = *pp;   // (the kernarg pointer intrinsic is null outside the kernel function itself)
  const Env<W> e = env_view<W>(p.recs + (size_t)env * p.L.rec_bytes, p.L);
  BBX_AS3 WideCold* const st = &ctl->st;
  int a;
  if (agent == BBX_AGENT_STDRANDOM) { uint32_t r = st->std_rng; a = std_choice(r, nP); if (lane_id() == 0) st->std_rng = r; }
  else a = select_pair<W>(e, nP, agent, [&](int g) { return (int)e.psug[g]; });
  if (lane_id() == 0) ctl->bc[0] = a;
}
template <int W>
__device__ __noinline__ void wide_leader_trace(const BbxParams* pp, BBX_AS3 WideCtl* ctl, int env, int nP, int nG, int nG_before, int action, int done, double reward) {
  const BbxParams& p = *pp;   // (the kernarg pointer intrinsic is null outside the kernel function itself)
  const Env<W> e = env_view<W>(p.recs + (size_t)env * p.L.rec_bytes, p.L);
  const uint64_t oh = wave_obs<W, true>(e, p, env, nP, false, true);
  const uint64_t ph = wave_pairs_hash<W, Env<W>>(e, nP);
  const uint64_t nh = nG > nG_before ? wave_poly_hash<W>(e, nG - 1) : 0;
  if (lane_id() == 0) {
    BbxTraceRec& tr = p.trace[(size_t)env * p.trace_stride + ctl->st.rollout_pos];
    tr.action = action; tr.nP = nP; tr.nG = nG; tr.done = done; tr.reward = reward;
    tr.obs_hash = oh; tr.pairs_hash = ph; tr.newpoly_hash = nh;
  }
}

template <int W, bool TRACE, bool LAZY, bool ACCT = !LAZY>
__device__ __forceinline__ void wide_body(char* smem) {
#ifdef BBX_PROF_BUILD
  unsigned long long wprof[6] = {0, 0, 0, 0, 0, 0}, wcnt[7] = {0, 0, 0, 0, 0, 0, 0};
  unsigned long long wlast = __builtin_amdgcn_s_memtime();
#endif
  WideCtx x;
  x.ctl = (BBX_AS3 WideCtl*)smem;
  x.tid = (int)threadIdx.x; x.lane = x.tid & (WAVE - 1); x.wave = uni(x.tid / WAVE);
  x.NT = uni((int)blockDim.x); x.NW = x.NT / WAVE;
  x.par_found = x.par_count = x.par_end = 0;
  int par_any = 0;                                         // (eager variants only: the run scan's exchange slot)
#ifdef BBX_PROF_BUILD
  x.prof_trips = 0;
  for (int i = 0; i < 5; i++) x.mt[i] = 0;
  x.ml = 0;
#endif
  const int env = (int)blockIdx.x;
  const bool leader = x.wave == 0;
  BBX_AS3 WideCold* const st = &x.ctl->st;

  // ---- LDS carve-up: [ctl][keys: H0 | H1 | F][coefficients: H0 | H1 | F][reducer table] ----------------------------
  int HC, FC, RC, SC, maxT;
  LdsKeys T;                                               // all term slots as one array: H0 | H1 | F | RB | S0 | S1
  WideTable<W> R;
  {
    const BbxParams& p = wide_params();
    HC = p.wide_hc; FC = p.wide_fc; RC = p.wide_rc; SC = LAZY ? p.wide_sc : 0; maxT = (int)p.L.maxT;
    BBX_AS3 char* q = (BBX_AS3 char*)smem + wide_ctl_bytes();
    const int nt = 2 * HC + FC + WRB + 2 * p.wide_sc;
    T.k = (BBX_AS3 uint64_t*)q; q += (size_t)nt * 8;
    T.c = (BBX_AS3 uint16_t*)q; q += ((size_t)nt * 2 + 15) & ~(size_t)15;
    R.lm = (BBX_AS3 uint32_t*)q; q += (size_t)RC * 4 * W;
    R.meta = (BBX_AS3 bbx_u32x2*)q; q += (size_t)RC * 8;
    R.poff = (BBX_AS3 uint32_t*)q;
  }
#define WIDE_REC(P) ((P).recs + (size_t)env * (P).L.rec_bytes)

  // ---- state: what the step loop branches on in registers, the rest in LDS ------------------------------------------
  int nG, nP, arena_used, status, need_reset, budget, t_agent;
  uint32_t agent_seed;
  {
    const BbxParams& p = wide_params();
    const BbxHdr* h = (const BbxHdr*)WIDE_REC(p);
    nG = uni(h->nG); nP = uni(h->nP); arena_used = uni(h->arena_used);
    status = uni(h->status); need_reset = uni(h->need_reset); budget = uni(h->budget); t_agent = uni(h->t);
    agent_seed = (uint32_t)uni((int)h->agent_seed);
    if (status == BBX_ST_STARVED || status == BBX_ST_SPILL || status == BBX_ST_TIMESLICE) status = BBX_ST_OK;
    if (p.set_budget) budget = bbx_st_capacity(status) ? budget + p.nsteps : p.nsteps;   // (bbx_common.h: bbx_st_capacity)
    if (p.pass == 1 && !(status == BBX_ST_OK && (need_reset || (budget > 0 && nP > 0)))) {         // (whole workgroup)
      if (p.wide_tail == 1 && x.tid == 0) atomicAdd(p.wide_done, 1);
      return;
    }
    if (x.tid == 0) {
      st->total_steps = h->total_steps; st->total_adds = h->total_additions; st->alg_bytes = h->alg_bytes;
      st->vret = h->vret; st->vdisc = h->vdisc; st->last_reward = 0.0;
      st->episode_steps = h->episode_steps; st->episodes = h->episodes; st->zero_red = h->zero_reductions; st->steps_done = 0;
      st->rollout_pos = h->rollout_pos; st->done_last = h->done_last; st->obs_trunc = h->obs_trunc; st->q_head = h->q_head;
      st->std_rng = h->std_rng; st->gen_state = h->gen_rng; st->rng_mark = h->std_rng;
      if (p.set_budget) { st->rollout_pos = 0; st->done_last = 0; st->vret = 0.0; st->vdisc = 1.0; st->obs_trunc = 0; }
    }
  }
  if (x.tid < 2 * WNWMAX) { (&x.ctl->wfound[0][0])[x.tid] = 0x7fffffff; (&x.ctl->wcount[0][0])[x.tid] = 0; (&x.ctl->wany[0][0])[x.tid] = 0; }
  if (x.tid < 2 * 4 * WNWMAX) (&x.ctl->wcnt4[0][0][0])[x.tid] = 0;
  __syncthreads();
  bool table_dirty = true;
  int rcl = 0;                                                                // reducers staged in LDS

  // ---- the polynomial being reduced --------------------------------------------------------------------------------
  // h = H + S.  H: live terms [hoff, hn) of LDS buffer H[cur] (in_lds) or of the record's HBM scratch buffer hbuf (0 / 1),
  // of which the LDS region H0 then caches the window [hw0, hw0 + hwn).  S (LAZY only): the accumulator, live terms
  // [soff, sn) of LDS buffer S[scur]: reducer tails are merged into S, H is rewritten only when S is full.  The lead
  // term of h is the larger of the two heads (their sum when the monomials agree; a zero sum is no term at all), so
  // the sequence of lead terms — and with it every reduction step, every remainder term, every reward — is exactly
  // that of the reference's eagerly merged h (buchberger.cpp:24-49).
  int hn = 0, hoff = 0, hsug = 0, cur = 0, hbuf = 0, hw0 = 0, hwn = 0;
  int sn = 0, soff = 0, scur = 0;
  bool in_lds = true;
#define WIDE_HM(P, E, B) ((E).hm + (size_t)(B) * (P).L.maxT)
#define WIDE_HC(P, E, B) ((E).hc + (size_t)(B) * (P).L.maxT)
#define WIDE_SACC(I) T.off(2 * HC + FC + WRB + (I) * SC)

  // Everything the loops below branch on is the same in every lane by construction; the compiler has to be told.  One value
  // it cannot prove uniform (a counter advanced under a branch it takes for per-lane) turns every loop-carried value after it
  // into a vector register and every branch into exec-mask code: the step loop then runs on spilled vector registers, with
  // reloads from scratch on the critical path of every reduction round.  WPIN breaks such chains at the loop heads.
#define WPIN(v) v = uni(v)
#define WPINB(v) v = uni((int)(v)) != 0

  // dst <- dst_live + B, where dst is S (to_s) or H and B is the live accumulator S (from_s; S is empty afterwards) or
  // (scale * x^shift) * f[0..fn), f = arena terms [foff, foff + fn)  (polynomials.cpp:148-202).  Returns false on
  // overflow (status set).  All threads, uniform arguments; ends with the result complete and visible.
  auto poly_add = [&](bool to_s, bool from_s, int foff, int fn, const Mono<W>& shift, uint32_t scale) __attribute__((always_inline)) -> bool {
    const BbxParams& p = wide_params();
    const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
    const Mono<W>* const fm = e.am + foff; const uint16_t* const fc = e.ac + foff;
    const LdsKeys F = T.off(2 * HC);                                          // the reducer-tail window
    const LdsKeys Sl = WIDE_SACC(scur).off(soff);                             // live accumulator
    const int sl = sn - soff;
    const int bn = from_s ? sl : fn;
    // tier 1 core: O[0..) <- A[0..an) + B, everything in LDS; an arena B streams through the window F in chunks
    auto lds_add = [&](const LdsKeys& A, int an, const LdsKeys& O, int ocap) __attribute__((always_inline)) -> int {
      if (from_s) return wide_merge(A, an, Sl, sl, O, 0, ocap, x);
      int nn = 0, a0 = 0, c0 = 0;
      do {
        WPIN(nn); WPIN(a0); WPIN(c0);
        const int cn = fn - c0 < FC ? fn - c0 : FC;
        WCOUNT(14, 1);
        wide_load_scaled<W>(F, fm + c0, fc + c0, cn, shift, scale, x);
        __syncthreads();
        WSTAMP(20);
        const int a1 = c0 + cn >= fn ? an : a0 + wide_count_ge(A.off(a0), an - a0, F.key(cn - 1));
        nn = wide_merge(A.off(a0), a1 - a0, F, cn, O, nn, ocap, x);
        WSTAMP(22);
        a0 = a1; c0 += cn;
      } while (c0 < fn);
      return nn;
    };
    if (to_s) {                                                               // S <- S + f  (the caller made sure it fits)
      WCOUNT(15, 1);
      sn = lds_add(Sl, sl, WIDE_SACC(scur ^ 1), SC);
      scur ^= 1; soff = 0;
      __syncthreads();
      return true;
    }
    int as = hoff;
    const int an = hn - as;
    if (bn == 0) return true;
    const bool keyable = wide_keys_ok<W>(hsug) || (W <= 4 && from_s);          // sugar bounds every degree in h and in x^shift * f
    if (an + bn > 2 * maxT) { status = BBX_ST_POLY_TOO_LONG; return false; }
    int nn = 0;
    if (keyable && an + bn <= HC) {
      // ---- tier 1: everything in LDS
      WCOUNT(10, 1); WCOUNT(16, an); WCOUNT(17, bn);
      if (!in_lds) {                                                          // H has shrunk back under the LDS capacity
        WCOUNT(18, 1);
        wide_load_plain<W>(T, WIDE_HM(p, e, hbuf) + as, WIDE_HC(p, e, hbuf) + as, an, x);
        as = 0; cur = 0; in_lds = true;
        __syncthreads();
      }
      WPIN(as); WPIN(cur); WPINB(in_lds);
      nn = lds_add(T.off(cur * HC + as), an, T.off((cur ^ 1) * HC), HC);
      cur ^= 1;
    } else {
      if (in_lds) {                                                           // H outgrows LDS: continue in the HBM scratch
        WCOUNT(19, 1);
        const LdsKeys Sx = T.off(cur * HC + as);
        Mono<W>* const dm = WIDE_HM(p, e, 0); uint16_t* const dc = WIDE_HC(p, e, 0);
        for (int t = x.tid; t < an; t += x.NT) { dm[t] = wide_unkey<W>(Sx.key(t)); dc[t] = (uint16_t)Sx.coef(t); }
        as = 0; in_lds = false; hbuf = 0;
        __syncthreads();
      }
      WPIN(as); WPINB(in_lds); WPIN(hbuf);
      const Mono<W>* const hm = WIDE_HM(p, e, hbuf) + as; const uint16_t* const hc = WIDE_HC(p, e, hbuf) + as;
      Mono<W>* const nm = WIDE_HM(p, e, hbuf ^ 1); uint16_t* const nc = WIDE_HC(p, e, hbuf ^ 1);
      const GlbOut<W> O{nm, nc};
      if (keyable && (from_s || fn <= HC + FC)) {
        // ---- tier 2: B entirely in LDS (the accumulator, or the scaled f staged in H1 | F), H streams through the
        // window H0, the result streams out
        WCOUNT(11, 1);
        const LdsKeys BB = from_s ? Sl : T.off(HC), Sw = T;
        if (!from_s) wide_load_scaled<W>(BB, fm, fc, fn, shift, scale, x);
        int b0 = 0, w0 = 0;
        do {
          WPIN(nn); WPIN(b0); WPIN(w0);
          const int wn = an - w0 < HC ? an - w0 : HC;
          wide_load_plain<W>(Sw, hm + w0, hc + w0, wn, x);
          __syncthreads();
          const int b1 = w0 + wn >= an ? bn : b0 + wide_count_ge(BB.off(b0), bn - b0, Sw.key(wn - 1));
          nn = wide_merge(Sw, wn, BB.off(b0), b1 - b0, O, nn, maxT, x);
          b0 = b1; w0 += wn;
        } while (w0 < an);
      } else {
        // ---- tier 3: the same merge on HBM-resident views (exponents beyond a byte, or f longer than the LDS)
        WCOUNT(12, 1);
        nn = uni(wide_merge_hbm<W>(hm, hc, an, fm, fc, fn, shift, scale, nm, nc, maxT, x));
        __syncthreads();
      }
      if (nn > maxT) { status = BBX_ST_POLY_TOO_LONG; return false; }
      hbuf ^= 1; hwn = 0;
    }
    hoff = 0; hn = nn;
    if (from_s) { sn = 0; soff = 0; }
    __syncthreads();                                                          // the new H is complete
    return true;
  };

  for (;;) {
    WSTAMP(8);
    if (status != BBX_ST_OK) break;
    if (need_reset) {                                                         // leader alone (once per episode), wave-level code
      if (leader) wide_leader_reset<W>(&wide_params(), x.ctl, env);
      __syncthreads();
      const int ok = uni(x.ctl->bc[0]);
      nG = uni(x.ctl->bc[1]); nP = uni(x.ctl->bc[2]); arena_used = uni(x.ctl->bc[3]); status = uni(x.ctl->bc[5]);
      __syncthreads();                                                        // bc[] may be rewritten from here on
      if (!ok) break;
      need_reset = 0; table_dirty = true;
    }
    if (budget <= 0) break;
    if (nP == 0) break;
    {
      // first kernel of a two-kernel launch (BbxParams::wide_tail): once the workgroups still at work would fit one per CU,
      // stop here, at a step boundary; the second kernel — no register cap, twice the LDS — takes the steps still owed
      const BbxParams& p = wide_params();
      if (p.wide_tail == 1) {
        if (x.tid == 0) x.ctl->bc[6] = __hip_atomic_load(p.wide_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const int left = p.B - uni(x.ctl->bc[6]);
        if (left <= p.wide_ncu) { status = BBX_ST_TIMESLICE; break; }
      }
    }
    if (x.tid == 0) st->rng_mark = st->std_rng;
    {
      const BbxParams& p = wide_params();
      if (nG + 1 > (int)p.L.maxG || nP - 1 + nG > (int)p.L.maxP || arena_used + maxT > (int)p.L.arena) {   // before anything is modified
        status = nG + 1 > (int)p.L.maxG ? BBX_ST_G_FULL : (nP - 1 + nG > (int)p.L.maxP ? BBX_ST_P_FULL : BBX_ST_ARENA_FULL);
        break;
      }
    }
    if (table_dirty) {                                                        // reducer table: lead monomial + metadata, reducer order
      const BbxParams& p = wide_params();
      const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
      rcl = nG < RC ? nG : RC;
      for (int r = x.tid; r < rcl; r += x.NT) {
        const int g = e.sidx[r];
        const Mono<W> s = e.slm[r];
        const uint32_t po = e.poff[g];
        bbx_u32x2 md; md.x = (uint32_t)e.plen[g] | ((uint32_t)e.psug[g] << 16); md.y = (uint32_t)e.pinv[g] | ((uint32_t)g << 16);
        R.put_mono(r, s); R.poff[r] = po; R.meta[r] = md;
      }
      table_dirty = false;
      __syncthreads();
    }

    WSTAMP(0);
    // ---- choose the pair, take it out of P, set up the S-polynomial -------------------------------------------------
    int action, sp_foff, sp_fn, sp_na;
    Mono<W> sp_shift; uint32_t sp_scale;
    {
      const BbxParams& p = wide_params();
      const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
      const int agent = p.agent;
      if (agent == BBX_AGENT_EXTERNAL) action = p.actions[env];
      else if (agent == BBX_AGENT_HASH) action = (int)bbx_agent_action32(agent_seed, (uint32_t)t_agent, (uint32_t)nP);
      else if (agent == BBX_AGENT_FIRST) action = 0;
      else if (agent == BBX_AGENT_LAST) action = nP - 1;
      else {                                                                  // seeded std random / the ordering strategies: leader
        if (leader) wide_leader_select<W>(&wide_params(), x.ctl, env, nP, agent);
        __syncthreads();
        action = x.ctl->bc[0];
        __syncthreads();
      }
      action = uni(action);
      if (action < 0 || action >= nP) { status = BBX_ST_BAD_ACTION; break; }
      const uint32_t pr = (uint32_t)uni((int)e.pairs[action]);
      const int gi = pr & 0xffffu, gj = pr >> 16;
      // (the pair leaves P — buchberger.cpp:319 — only once the reduction is through: until then the step works in LDS,
      // the scratch buffers and the free end of the arena, so a capacity miss leaves the record as the step found it and
      // the step is taken again after the host has enlarged the record: bbx_common.h, bbx_st_capacity)

      // S-polynomial  buchberger.cpp:18-21: h <- (gamma / LT g_i) tail(g_i); the loop below subtracts (gamma / LT g_j) tail(g_j)
      const Mono<W> lmi = e.lm[gi], lmj = e.lm[gj];
      const Mono<W> gamma = m_lcm(lmi, lmj);
      const int offi = uni((int)e.poff[gi]), offj = uni((int)e.poff[gj]);
      const int na = uni((int)e.plen[gi]) - 1, nb = uni((int)e.plen[gj]) - 1;
      Mono<W> shi = m_div(gamma, lmi), shj = m_div(gamma, lmj);
#pragma unroll
      for (int q = 0; q < W; q++) { shi.w[q] = (uint32_t)uni((int)shi.w[q]); shj.w[q] = (uint32_t)uni((int)shj.w[q]); }
      const uint32_t sci = (uint32_t)uni((int)e.pinv[gi]), scj = negmod((uint32_t)uni((int)e.pinv[gj]));
      const int si = uni((int)e.psug[gi]) + (int)m_deg(shi), sj = uni((int)e.psug[gj]) + (int)m_deg(shj);
      hsug = uni(si > sj ? si : sj);
      if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; break; }
      if (na + nb > 2 * maxT || na > maxT) { status = BBX_ST_POLY_TOO_LONG; break; }
      if (wide_keys_ok<W>(hsug) && na <= HC) {
        wide_load_scaled<W>(T, e.am + offi + 1, e.ac + offi + 1, na, shi, sci, x);
        in_lds = true; cur = 0;
      } else {
        Mono<W>* const dm = WIDE_HM(p, e, 1); uint16_t* const dc = WIDE_HC(p, e, 1);
        for (int t = x.tid; t < na; t += x.NT) { dm[t] = m_mul(e.am[offi + 1 + t], shi); dc[t] = (uint16_t)mulmod(e.ac[offi + 1 + t], sci); }
        in_lds = false; hbuf = 1;
      }
      WPINB(in_lds); WPIN(cur); WPIN(hbuf);
      hoff = 0; hn = na; hwn = 0;
      __syncthreads();
      sp_foff = offj + 1; sp_fn = nb; sp_shift = shj; sp_scale = scj; sp_na = na;
    }

    WSTAMP(1);
    // ---- reduce  buchberger.cpp:24-49 (its first trip subtracts the second half of the S-polynomial) ---------------
    int nsteps_red = 0, rn = 0, rsug = 0, rflushed = 0;
    long long step_bytes = 0;
    bool overflow = false, first = true;
    sn = 0; soff = 0;
    const LdsKeys RB = T.off(2 * HC + FC);
    auto flush_r = [&]() __attribute__((always_inline)) {                     // buffered terms [rflushed, rn) of r -> arena (all threads)
      __syncthreads();
      if (rn > rflushed) {
        const BbxParams& p = wide_params();
        const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
        for (int t = x.tid; t < rn - rflushed; t += x.NT) {
          e.am[arena_used + rflushed + t] = wide_unkey<W>(RB.key(t)); e.ac[arena_used + rflushed + t] = (uint16_t)RB.coef(t);
        }
      }
      rflushed = rn;
      __syncthreads();
    };
    for (;;) {
      int fn, foff, found = -1, an = 0;
      Mono<W> shift; uint32_t scale;
      if (first) { foff = sp_foff; fn = sp_fn; shift = sp_shift; scale = sp_scale; }
      else {
        // ---- the lead term of h = H + S leaves h (it is either cancelled by the reducer or moved to r)
        Mono<W> lmh; uint32_t lch = 0;
        bool zero = false;
        if (W <= 4 && in_lds && wide_keys_ok<W>(hsug)) {
          // the common case — H in LDS as keys — without a branch per case: both heads come in one trip to LDS, an exhausted
          // side reads as key 0 (below every key), the larger head leaves h (both when the monomials agree)
          const LdsKeys Hc = T.off(cur * HC), Sc = WIDE_SACC(scur);
          for (;;) {
            const bool hH = hoff < hn, hS = LAZY && soff < sn;
            if (!hH && !hS) { zero = true; break; }
            uint64_t kH = Hc.key(hH ? hoff : 0), kS = LAZY ? Sc.key(hS ? soff : 0) : 0;
            uint32_t cH = Hc.coef(hH ? hoff : 0), cS = LAZY ? Sc.coef(hS ? soff : 0) : 0;
            kH = hH ? uni64(kH) : 0; kS = hS ? uni64(kS) : 0; cH = uni(cH); cS = uni(cS);
            const bool fromH = kH >= kS, fromS = kS >= kH;
            lch = addmod(fromH ? cH : 0u, fromS ? cS : 0u);
            hoff += fromH ? 1 : 0; soff += fromS ? 1 : 0;
            if (lch != 0) { lmh = wide_unkey<W>(fromH ? kH : kS); break; }   // (a zero sum: the term does not exist)
          }
        } else
        for (;;) {
          const bool hH = hoff < hn, hS = LAZY && soff < sn;
          if (!hH && !hS) { zero = true; break; }
          uint64_t kH = 0, kS = 0; uint32_t cH = 0, cS = 0;
          Mono<W> mH = m_zero<W>();
          const bool keyed = W <= 4 && (in_lds || wide_keys_ok<W>(hsug));     // H terms comparable as keys
          if (hH) {
            if (in_lds) { const LdsKeys Hc = T.off(cur * HC); kH = Hc.key(hoff); cH = Hc.coef(hoff); }
            else {
              if (hoff < hw0 || hoff >= hw0 + hwn) {                          // refill the LDS window on the HBM-resident H
                const BbxParams& p = wide_params();
                const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
                __syncthreads();
                hw0 = hoff; hwn = hn - hoff < HC ? hn - hoff : HC;
                if (keyed) wide_load_plain<W>(T, WIDE_HM(p, e, hbuf) + hw0, WIDE_HC(p, e, hbuf) + hw0, hwn, x);
                else { hwn = 1; mH = WIDE_HM(p, e, hbuf)[hoff]; cH = WIDE_HC(p, e, hbuf)[hoff]; }
                __syncthreads();
              }
              WPIN(hw0); WPIN(hwn);
              if (keyed) { kH = T.key(hoff - hw0); cH = T.coef(hoff - hw0); }
            }
          }
          if (hS) { const LdsKeys Sc = WIDE_SACC(scur); kS = Sc.key(soff); cS = Sc.coef(soff); }
          kH = uni64(kH); kS = uni64(kS); cH = (uint32_t)uni((int)cH); cS = (uint32_t)uni((int)cS);
          if (!keyed) { lmh = mH; lch = cH; hoff++; hwn = 0; break; }          // (then S is empty: see poly_add)
          if (hH && (!hS || kH > kS)) { lmh = wide_unkey<W>(kH); lch = cH; hoff++; break; }
          if (hS && (!hH || kS > kH)) { lmh = wide_unkey<W>(kS); lch = cS; soff++; break; }
          hoff++; soff++;                                                     // the same monomial in both
          lch = addmod(cH, cS);
          if (lch != 0) { lmh = wide_unkey<W>(kH); break; }                   // (a zero sum: the term does not exist)
        }
        if (zero) break;
#pragma unroll
        for (int q = 0; q < W; q++) lmh.w[q] = (uint32_t)uni((int)lmh.w[q]);
        lch = (uint32_t)uni((int)lch);
        const Mono<W>* slm_g = nullptr;                                       // (reducers beyond the LDS table: rare)
        if (nG > rcl) { const BbxParams& p = wide_params(); slm_g = (const Mono<W>*)(WIDE_REC(p) + p.L.off_slm); }
        found = wide_find_divisor<W>(R, rcl, slm_g, nG, lmh, x);
        WSTAMP(2);
#ifdef BBX_PROF_BUILD
        WCOUNT(23, x.prof_trips); x.prof_trips = 0; WCOUNT(24, found >= 0 ? found : 0); WCOUNT(25, nG);
#endif
        an = (hn - hoff) + (sn - soff);                                       // (exact in the eager variant: S is empty)
        if (found < 0) {                                                      // r <- r + LT h ; h <- h - LT h   (41-44)
          if (rn >= maxT) { status = BBX_ST_POLY_TOO_LONG; overflow = true; break; }
          step_bytes += 8LL * nG + 12LL * (2 * (an + 1) - 1);
          // r grows directly behind the arena's end.  Its terms collect in LDS and go out WRB at a time: a store per
          // term would stall the next barrier's s_waitcnt vmcnt(0) for a round trip to memory
          if (wide_keys_ok<W>(hsug)) {
            if (rn - rflushed == WRB) flush_r();
            if (x.tid == 0) RB.put(rn - rflushed, wide_key(lmh), lch);
          } else {
            flush_r();
            if (x.tid == 0) {
              const BbxParams& p = wide_params();
              const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
              e.am[arena_used + rn] = lmh; e.ac[arena_used + rn] = (uint16_t)lch;
            }
            rflushed = rn + 1;
          }
          WPIN(rflushed);
          const int d = (int)m_deg(lmh);
          rsug = d > rsug ? d : rsug;
          rn++;
          WCOUNT(13, 1);
          // A run of irreducible terms (eager variants: under the ordering strategies a term that goes to r is usually
          // followed by more; with the accumulator variant's random-agent workloads the extra scan was measured to cost
          // 15 %): while the next terms of h sit in LDS as keys, four of them are tested per scan; the leading irreducible
          // ones move to r together, the first reducible one is left to the next round.
          while (!LAZY && in_lds && wide_keys_ok<W>(hsug) && hn - hoff >= 2 && rn + 4 <= maxT) {
            const LdsKeys Hc = T.off(cur * HC);
            const int nc = hn - hoff < 4 ? hn - hoff : 4;
            Mono<W> cm[4]; uint64_t ck[4]; uint32_t cc[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int at = hoff + (j < nc ? j : 0);
              ck[j] = Hc.key(at); cc[j] = Hc.coef(at);
              cm[j] = wide_unkey<W>(ck[j]);
#pragma unroll
              for (int q = 0; q < W; q++) cm[j].w[q] = (uint32_t)uni((int)cm[j].w[q]);
            }
            const int red = wide_reducible_mask<W>(R, rcl, slm_g, nG, cm, nc, x, par_any);
            int mv = red ? __builtin_ctz((unsigned)red) : 4;                  // leading irreducible candidates
            mv = mv < nc ? mv : nc;
            if (mv == 0) break;
            if (rn - rflushed + mv > WRB) flush_r();
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (j < mv) {
                if (x.tid == 0) RB.put(rn - rflushed + j, ck[j], cc[j]);
                const int dj = (int)m_deg(cm[j]);
                rsug = dj > rsug ? dj : rsug;
                step_bytes += 8LL * nG + 12LL * (2 * ((hn - hoff - 1 - j) + 1) - 1);
              }
            }
            rn += mv; hoff += mv;
            WCOUNT(13, mv);
            if (mv < nc) break;
          }
          WSTAMP(3);
          continue;
        }
        // h <- h - (LT h / LT f) f     (34-36)
        Mono<W> lmg; uint32_t poffg, md0, md1;
        if (found < rcl) { lmg = R.mono(found); poffg = R.poff[found]; const bbx_u32x2 md = R.meta[found]; md0 = md.x; md1 = md.y; }
        else {
          const BbxParams& p = wide_params();
          const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
          const int g = e.sidx[found];
          lmg = e.slm[found]; poffg = e.poff[g];
          md0 = (uint32_t)e.plen[g] | ((uint32_t)e.psug[g] << 16); md1 = (uint32_t)e.pinv[g] | ((uint32_t)g << 16);
        }
        poffg = (uint32_t)uni((int)poffg); md0 = (uint32_t)uni((int)md0); md1 = (uint32_t)uni((int)md1);
#pragma unroll
        for (int q = 0; q < W; q++) lmg.w[q] = (uint32_t)uni((int)lmg.w[q]);
        fn = (int)(md0 & 0xffffu) - 1;
        shift = m_div(lmh, lmg);
        scale = negmod(mulmod(lch, md1 & 0xffffu));
        const int fs = (int)(md0 >> 16) + (int)m_deg(shift);
        hsug = uni(fs > hsug ? fs : hsug);
        if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; overflow = true; break; }
        foff = (int)poffg + 1;
      }
      WSTAMP(4);
      // where the scaled reducer tail goes: into the accumulator while it fits there (and exponents fit a byte), else
      // into H — after the accumulator has been emptied into H (its terms are all byte-sized: they were when they went in)
      // — and only while that saves work: a short H (one merge tile with the tail) is rewritten just as cheaply
      const bool to_s = LAZY && fn > 0 && fn <= SC && (W == 2 || hsug <= 255) &&   // (written out, not wide_keys_ok: as a call the
                        // condition lands in other blocks and the lazy kernel spills 152 instead of 60 registers — 18.6 k instead of
                        // 33.8 k env-steps/s on cyclic-7; 32-byte monomials never run the lazy variant)
                        (sn - soff > 0 || !in_lds || (hn - hoff) + fn > x.NT * WSEG);
      const bool flush_s = LAZY && sn - soff > 0 && (!to_s || (sn - soff) + fn > SC);
      bool ok = true;
      for (int pass = flush_s ? 0 : 1; pass < 2 && ok; pass++) ok = uni((int)poly_add(pass == 1 && to_s, pass == 0, foff, fn, shift, scale)) != 0;
      if (!ok) { overflow = true; break; }
      if (in_lds) WSTAMP(5); else WSTAMP(6);
      if (first) { step_bytes += 12LL * (sp_na + fn + 2 + (hn - hoff)); first = false; }
      else {
        step_bytes += 8LL * (found + 1) + 12LL * (fn + 1) + 12LL * (an + 1 + (hn - hoff));
        nsteps_red++;
        if (nsteps_red > (1 << 24)) { status = BBX_ST_RUNAWAY; overflow = true; break; }
      }
    }
    if (overflow) break;
    if (rn > 65535) { status = BBX_ST_POLY_LIMIT; break; }                    // plen[] is 16 bits
    flush_r();
    rsug = rsug > hsug ? rsug : hsug;                                         // sugar of r + h (48)
    {
      // P.erase(remove(action))  buchberger.cpp:319 — stable, all threads (flush_r's barrier orders every earlier read of
      // the pair list before the first write); from here on the step cannot fail for capacity
      const BbxParams& p = wide_params();
      const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
      for (int base = action; base < nP - 1; base += x.NT) {
        const int k = base + x.tid;
        uint32_t v = 0;
        if (k < nP - 1) v = e.pairs[k + 1];
        __syncthreads();
        if (k < nP - 1) e.pairs[k] = v;
      }
      nP -= 1;
      __syncthreads();                                                        // the list is complete for whoever reads it next
    }
    WSTAMP(3);

    // ---- basis / pair-set update  buchberger.cpp:321-327: leader, wave-level code on the HBM record ----------------
    const int nG_before = nG, nP_before = nP;
    if (rn != 0) {
      if (leader) wide_leader_add<W>(&wide_params(), x.ctl, env, nG, nP, arena_used, rn, rsug);
      __syncthreads();
      const int ok = uni(x.ctl->bc[0]);
      nG = uni(x.ctl->bc[1]); nP = uni(x.ctl->bc[2]); arena_used = uni(x.ctl->bc[3]); status = uni(x.ctl->bc[5]);
      __syncthreads();
      if (!ok) break;
      table_dirty = true;
      step_bytes += 12LL * rn + 8LL * nG_before + 8LL * (nP_before + nP);
    }
    const bool done = nP == 0;
    WSTAMP(7);
    {
      const BbxParams& p = wide_params();
      step_bytes += 4LL * nP * 2 * p.nvars * p.k;
      const double reward = (p.rewards_mode == BBX_REW_ADDITIONS) ? (-1.0 - (double)nsteps_red) : -1.0;  // 328
      if (p.obs_every_step && p.obs) {
        const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
        wide_obs<W>(e, p, env, nP, x);
      }
      if (TRACE && p.trace != nullptr && leader) wide_leader_trace<W>(&wide_params(), x.ctl, env, nP, nG, nG_before, action, done ? 1 : 0, reward);
      if (x.tid == 0) {                                                       // bookkeeping (single writer)
        if (ACCT && !LAZY) st->alg_bytes += step_bytes;                        // (the accumulator hides the canonical length of h)
        st->last_reward = reward;
        if (p.value_mode) { double vr = st->vret, vd = st->vdisc; value_accumulate(vr, vd, reward, p.gamma); st->vret = vr; st->vdisc = vd; }
        st->total_steps += 1; st->total_adds += 1 + nsteps_red; st->episode_steps += 1; st->steps_done += 1;
        if (rn == 0) st->zero_red += 1;
        if (p.obs_every_step && p.obs && nP > p.obs_rows) st->obs_trunc = 1;
        st->rollout_pos += 1; st->done_last = done ? 1 : 0;
        if (done) st->episodes += 1;
      }
      t_agent++; budget--;
      if (done && p.auto_reset) need_reset = 1;
    }
  }

  __syncthreads();
#ifdef BBX_PROF_BUILD
  if (x.tid == 0) {
    const int tslot[6] = {2, 4, 20, 22, 5, 8}, cslot[7] = {10, 13, 23, 24, 25, 15, 31};
    for (int i = 0; i < 6; i++) atomicAdd(&bbx_wide_prof_acc[tslot[i]], wprof[i]);
    for (int i = 0; i < 7; i++) atomicAdd(&bbx_wide_prof_acc[cslot[i]], wcnt[i]);
    for (int i = 0; i < 5; i++) atomicAdd(&bbx_wide_prof_acc[26 + i], x.mt[i]);
  }
#endif
  {
    const BbxParams& p = wide_params();
    const Env<W> e = env_view<W>(WIDE_REC(p), p.L);
    if (p.obs && status == BBX_ST_OK) wide_obs<W>(e, p, env, nP, x);
    if (x.tid == 0) {
      BbxHdr* h = (BbxHdr*)WIDE_REC(p);
      const int obs_trunc = st->obs_trunc | ((p.obs && status == BBX_ST_OK && nP > p.obs_rows) ? 1 : 0);
      const int steps_done = st->steps_done, done_last = st->done_last, q_head = st->q_head;
      h->nG = nG; h->nP = nP; h->arena_used = arena_used; h->status = status; h->need_reset = need_reset;
      h->q_head = q_head; h->t = t_agent; h->std_rng = bbx_st_capacity(status) ? st->rng_mark : st->std_rng; h->gen_rng = st->gen_state; h->episode_steps = st->episode_steps;
      h->total_steps = st->total_steps; h->total_additions = st->total_adds; h->episodes = st->episodes; h->zero_reductions = st->zero_red;
      h->steps_done = steps_done; h->budget = budget; h->rollout_pos = st->rollout_pos; h->done_last = done_last; h->alg_bytes = st->alg_bytes;
      h->vret = st->vret; h->vdisc = st->vdisc; h->obs_trunc = obs_trunc;
      if (p.lite) *(int4*)(p.lite + 4 * (size_t)env) = make_int4(status | (obs_trunc ? BBX_LITE_OBS_TRUNC : 0), q_head, budget, nP);
      if (p.value_mode && p.values) p.values[env] = st->vret;
      if (p.rewards && (steps_done > 0 || p.pass == 0)) p.rewards[env] = st->last_reward;
      if (p.dones) p.dones[env] = (uint8_t)((done_last || (nP == 0 && !need_reset)) ? 1 : 0);
      if (p.rows) p.rows[env] = nP;
      if (p.wide_tail == 1) atomicAdd(p.wide_done, 1);
    }
  }
#undef WPIN
#undef WPINB
#undef WIDE_REC
#undef WIDE_HM
#undef WIDE_HC
#undef WIDE_SACC
}

// LAZY = false: h eagerly merged every round — with ACCT its exact length enters the algorithmic bytes (the accounting
// variant), without it this is the lean kernel of the ordering strategies (First / Degree / Normal / Sugar and their
// reversals: runs in which h stays short, where the accumulator's bookkeeping is pure overhead: 6.2 s instead of 7.4 s
// for the single cyclic-7 run to completion).
// LAZY = true: the lean variant with the accumulator (random and external agents: h grows long; 2x at cyclic-7 random).
template <int W, bool TRACE, bool LAZY>
__global__ __launch_bounds__(512, 4) void bbx_wide_kernel(BbxParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wide_body<W, TRACE, LAZY>(smem);
}
template <int W>
__global__ __launch_bounds__(512, 4) void bbx_wide_eager_kernel(BbxParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wide_body<W, false, false, false>(smem);
}
// the same code for batches of at most one workgroup per CU (two waves per SIMD): no 128-register cap, no spills
template <int W, bool LAZY, bool ACCT>
__global__ __launch_bounds__(512, 2) void bbx_wide_kernel_1cu(BbxParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  wide_body<W, false, LAZY, ACCT>(smem);
}
