// Device code shared by the kernel classes of libbbx (gfx950 / MI355X): included by bbx_general.hip, bbx_binom.hip,
// bbx_fast.hip, bbx_wide.hip and bbx_aux.hip.  Everything here is a template or an inline device function.
//
// One 64-lane wavefront owns one environment for the whole launch and runs
// `nsteps` environment steps back to back:
//
//   select pair -> S-polynomial -> full reduction -> Gebauer-Moeller update -> reducer insert -> observation
//
// replacing, bit for bit, the reference's
//   LeadMonomialsEnv::step / BuchbergerEnv::step   deepgroebner/buchberger.cpp:318-329, 398-408
//   spoly / reduce / update                        deepgroebner/buchberger.cpp:18-99
//   Polynomial +,-,Term*  and Monomial ops         deepgroebner/polynomials.cpp:41-202
//   BuchbergerEnv::reset (from a host-generated ideal) deepgroebner/buchberger.cpp:299-315
//
// Design notes (MI355X):
//  * integer/indexing work, no MFMA.  Exponent vectors are packed u16 pairs so that lcm / product /
//    quotient / divisibility are v_pk_max_u16 / v_pk_add_u16 / v_pk_sub_u16 [clamp] on whole words.
//  * the first-divisor scan reads the reducers' lead monomials in reducer order, lane k <- slm[k]
//    (coalesced 8/16-B loads), tests divisibility per lane, and takes the first set bit of the
//    64-bit ballot — identical to the reference's linear scan with `break`.
//  * kernel template STAGED keeps the whole environment record in LDS for the launch (small
//    classes: 3-variable binomial ideals need ~6 KB); otherwise the record is worked on in HBM/L2.
//  * no inter-workgroup communication at all: environments are independent.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdint.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "bbx_common.h"

#define WAVE 64

// ------------------------------------------------------------------ wave helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }
__device__ __forceinline__ void wave_sync() {
  // lanes of one wave exchange data through LDS / their own HBM record; memory operations of a
  // wave complete in order, so a wavefront-scope fence (a compiler barrier, no cache action) is all
  // that is needed between a write by one lane and a read by another.
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }   // (not __ballot: that one materialises the predicate in a VGPR first)
__device__ __forceinline__ int prefix_of(uint64_t mask, int lane) { return __popcll(mask & ((1ull << lane) - 1ull)); }
__device__ __forceinline__ int uni(int x) { return __builtin_amdgcn_readfirstlane(x); }
// A value every lane of the wave holds alike, pinned to scalar registers.  The compiler proves uniformity from the data
// flow; one counter advanced under a branch it takes for per-lane, one value loaded from LDS, and every loop-carried
// value downstream becomes a vector register and every branch on it exec-mask code.  The step loops pin their state at
// the loop heads (pin(x)): a v_readfirstlane per value where the proof fails, nothing where it holds.
__device__ __forceinline__ uint32_t uni(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
__device__ __forceinline__ bool uni(bool x) { return __builtin_amdgcn_readfirstlane((int)x) != 0; }
__device__ __forceinline__ uint64_t uni(uint64_t x) { return ((uint64_t)uni((uint32_t)(x >> 32)) << 32) | uni((uint32_t)x); }
__device__ __forceinline__ long long uni(long long x) { return (long long)uni((uint64_t)x); }
__device__ __forceinline__ double uni(double x) { return __longlong_as_double(uni(__double_as_longlong(x))); }
template <class T> __device__ __forceinline__ void pin(T& v) { v = uni(v); }
template <class T, class... R> __device__ __forceinline__ void pin(T& v, R&... r) { v = uni(v); pin(r...); }
// Load from memory that no kernel writes (the ideal queue: filled by the host between launches) through the constant
// address space: with a wave-uniform address this is a scalar load, the value lives in an SGPR and everything derived
// from it (loop bounds, branch conditions) stays on the scalar unit.
typedef const __attribute__((address_space(4))) uint32_t* bbx_cptr32;
__device__ __forceinline__ uint32_t ldc(const uint32_t* q) { return *(bbx_cptr32)(uintptr_t)q; }
__device__ __forceinline__ int ldc(const int32_t* q) { return (int)*(bbx_cptr32)(uintptr_t)q; }
__device__ __forceinline__ uint64_t wave_sum64(uint64_t v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
  return v;
}
__device__ __forceinline__ uint64_t wave_min64(uint64_t v) {
  for (int o = 32; o > 0; o >>= 1) { uint64_t w = __shfl_xor(v, o, WAVE); v = w < v ? w : v; }
  return v;
}

// wave-wide unsigned minimum with DPP row shifts / row broadcasts (no LDS traffic); result is uniform
__device__ __forceinline__ uint32_t wave_min32(uint32_t x) {
#define BBX_DPPMIN(ctrl, rmask) { uint32_t y_ = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, ctrl, rmask, 0xF, false); x = y_ < x ? y_ : x; }
  BBX_DPPMIN(0x111, 0xF) BBX_DPPMIN(0x112, 0xF) BBX_DPPMIN(0x114, 0xF) BBX_DPPMIN(0x118, 0xF) BBX_DPPMIN(0x142, 0xA) BBX_DPPMIN(0x143, 0xC)
#undef BBX_DPPMIN
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}

// ------------------------------------------------------------------ GF(32003)   polynomials.h:10-26
// a, b in [0, P): the product is below 2^30, where floor(x / P) == (x * ceil(2^45 / P)) >> 45 exactly (the error term
// x * (M*P - 2^45) / (P * 2^45) stays below 2^-15 < 1/P), i.e. one high multiply instead of the generic 33-bit magic
__device__ __forceinline__ uint32_t mulmod(uint32_t a, uint32_t b) {
  const uint32_t x = a * b;
  const uint32_t q = (uint32_t)(((uint64_t)x * 1099408559ull) >> 45);
  return x - q * BBX_P;
}
__device__ __forceinline__ uint32_t addmod(uint32_t a, uint32_t b) { uint32_t s = a + b; return s >= BBX_P ? s - BBX_P : s; }
__device__ __forceinline__ uint32_t negmod(uint32_t a) { return a ? BBX_P - a : 0u; }
// inverse (polynomials.cpp:11-23 computes the same unique field element by extended Euclid):
// a^(P-2), P-2 = 32001 = 0b111110100000001
__device__ inline uint32_t invmod(uint32_t a) {
  uint32_t r = 1, b = a;
  uint32_t e = BBX_P - 2;
#pragma unroll 1
  while (e) { if (e & 1) r = mulmod(r, b); b = mulmod(b, b); e >>= 1; }
  return r;
}

// ------------------------------------------------------------------ packed monomials   polynomials.h:29-55
typedef unsigned short us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b))); }
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b))); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b)); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b)); }
__device__ __forceinline__ uint32_t pk_subsat(uint32_t a, uint32_t b) { return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b))); }

template <int W> struct __attribute__((aligned(W * 4))) Mono { uint32_t w[W]; };

// a monomial to / from memory word by word (adjacent words merge into one wide access): a struct assignment between address
// spaces is a memcpy through a stack slot when it stands under a condition, and the slot then stays in scratch memory
template <int W> __device__ __forceinline__ Mono<W> m_ld(const Mono<W>* p) {
  Mono<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = p->w[i];
  return r;
}
template <int W> __device__ __forceinline__ void m_st(Mono<W>* p, const Mono<W>& v) {
#pragma unroll
  for (int i = 0; i < W; i++) p->w[i] = v.w[i];
}
template <int W> __device__ __forceinline__ uint32_t m_deg(const Mono<W>& a) { return a.w[W - 1] >> 16; }
template <int W> __device__ __forceinline__ Mono<W> m_zero() { Mono<W> r; for (int i = 0; i < W; i++) r.w[i] = 0; return r; }
template <int W> __device__ __forceinline__ Mono<W> m_mul(const Mono<W>& a, const Mono<W>& b) {  // cpp:41-47 (degree slot adds too)
  Mono<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = pk_add(a.w[i], b.w[i]);
  return r;
}
template <int W> __device__ __forceinline__ Mono<W> m_div(const Mono<W>& a, const Mono<W>& b) {  // cpp:50-57
  Mono<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = pk_sub(a.w[i], b.w[i]);
  return r;
}
template <int W> __device__ __forceinline__ Mono<W> m_lcm(const Mono<W>& a, const Mono<W>& b) {  // cpp:111-118
  Mono<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = pk_max(a.w[i], b.w[i]);
  // recompute the degree slot = sum of the exponent slots
  uint32_t t = 0;
#pragma unroll
  for (int i = 0; i < W - 1; i++) t += r.w[i];            // halves add independently (sums < 65536)
  uint32_t d = (t & 0xffffu) + (t >> 16) + (r.w[W - 1] & 0xffffu);
  r.w[W - 1] = (r.w[W - 1] & 0xffffu) | (d << 16);
  return r;
}
template <int W> __device__ __forceinline__ bool m_eq(const Mono<W>& a, const Mono<W>& b) {  // cpp:77-81
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < W; i++) x |= a.w[i] ^ b.w[i];
  return x == 0;
}
// a | b  (is_divisible(b, a), cpp:93-98): every exponent of a <= that of b
template <int W> __device__ __forceinline__ bool m_divides(const Mono<W>& a, const Mono<W>& b) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < W; i++) x |= pk_subsat(a.w[i], b.w[i]);
  return x == 0;
}
// gcd == 1  <=>  lcm(a,b) == a*b, the test at buchberger.cpp:65 and :88
template <int W> __device__ __forceinline__ bool m_coprime(const Mono<W>& a, const Mono<W>& b) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < W - 1; i++) x |= pk_min(a.w[i], b.w[i]);
  x |= pk_min(a.w[W - 1], b.w[W - 1]) & 0xffffu;
  return x == 0;
}
// grevlex a > b (cpp:60-74): degree first, then from the last variable down the SMALLER exponent wins.
// With the degree in the most significant slot and the exponent slots complemented this is one
// unsigned compare of the whole monomial.
__device__ __forceinline__ bool m_gt(const Mono<2>& a, const Mono<2>& b) {
  uint64_t ka = (((uint64_t)a.w[1] << 32) | a.w[0]) ^ 0x0000FFFFFFFFFFFFull;
  uint64_t kb = (((uint64_t)b.w[1] << 32) | b.w[0]) ^ 0x0000FFFFFFFFFFFFull;
  return ka > kb;
}
__device__ __forceinline__ bool m_gt(const Mono<4>& a, const Mono<4>& b) {
  uint64_t ha = (((uint64_t)a.w[3] << 32) | a.w[2]) ^ 0x0000FFFFFFFFFFFFull;
  uint64_t hb = (((uint64_t)b.w[3] << 32) | b.w[2]) ^ 0x0000FFFFFFFFFFFFull;
  uint64_t la = ~(((uint64_t)a.w[1] << 32) | a.w[0]);
  uint64_t lb = ~(((uint64_t)b.w[1] << 32) | b.w[0]);
  return ha > hb || (ha == hb && la > lb);
}
// 8-variable rings (the reference's N = 8, polynomials.h:29): 32-byte monomials, 15 exponent slots + degree.  Most
// significant word first; the degree is the high half of the last word, every exponent slot is complemented.
__device__ __forceinline__ bool m_gt(const Mono<8>& a, const Mono<8>& b) {
#pragma unroll
  for (int i = 7; i >= 0; i--) {
    const uint32_t mk = i == 7 ? 0x0000FFFFu : 0xFFFFFFFFu;
    const uint32_t ka = a.w[i] ^ mk, kb = b.w[i] ^ mk;
    if (ka != kb) return ka > kb;
  }
  return false;
}
template <int W> __device__ __forceinline__ uint32_t m_exp(const Mono<W>& a, int v) {
  uint32_t w = a.w[0];                          // select chain, not a runtime index (keeps Mono in VGPRs)
#pragma unroll
  for (int i = 1; i < W; i++) w = ((v >> 1) == i) ? a.w[i] : w;
  return (v & 1) ? (w >> 16) : (w & 0xffffu);
}

// the n exponents of one monomial slot of the observation matrix, written with the widest stores 4-byte alignment
// allows (dwordx4 / dwordx2 / dword) instead of n single dwords: the observation is the largest output of a step
struct __attribute__((aligned(4))) ObsI4 { int32_t a, b, c, d; };
struct __attribute__((aligned(4))) ObsI2 { int32_t a, b; };
// Observation rows are written once and read by a later kernel (or the host): non-temporal stores, so that the stream
// of rows (tens of MB per batch step in the HBM-resident classes) does not push the environments' records out of the
// caches the gathers of the next step hit.  BBX_OBS_TEMPORAL restores plain stores (A/B experiments).
typedef int32_t bbx_obs_i4 __attribute__((ext_vector_type(4), aligned(4)));
typedef int32_t bbx_obs_i2 __attribute__((ext_vector_type(2), aligned(4)));
#ifdef BBX_OBS_TEMPORAL
#define BBX_OBS_ST(P, V) (*(P) = (V))
#else
#define BBX_OBS_ST(P, V) __builtin_nontemporal_store((V), (P))
#endif
template <int W> __device__ __forceinline__ void obs_store(int32_t* dst, const Mono<W>& mm, int n) {
  int32_t x[2 * W > 8 ? 2 * W : 8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < W; i++) { x[2 * i] = (int32_t)(mm.w[i] & 0xffffu); x[2 * i + 1] = (int32_t)(mm.w[i] >> 16); }
  const bbx_obs_i4 lo = {x[0], x[1], x[2], x[3]}, hi = {x[4], x[5], x[6], x[7]};
  const bbx_obs_i2 l2 = {x[0], x[1]}, h2 = {x[4], x[5]};
  switch (n) {                                             // wave-uniform
    case 1: BBX_OBS_ST(dst, x[0]); break;
    case 2: BBX_OBS_ST((bbx_obs_i2*)dst, l2); break;
    case 3: BBX_OBS_ST((bbx_obs_i2*)dst, l2); BBX_OBS_ST(dst + 2, x[2]); break;
    case 4: BBX_OBS_ST((bbx_obs_i4*)dst, lo); break;
    case 5: BBX_OBS_ST((bbx_obs_i4*)dst, lo); BBX_OBS_ST(dst + 4, x[4]); break;
    case 6: BBX_OBS_ST((bbx_obs_i4*)dst, lo); BBX_OBS_ST((bbx_obs_i2*)(dst + 4), h2); break;
    case 7: BBX_OBS_ST((bbx_obs_i4*)dst, lo); BBX_OBS_ST((bbx_obs_i2*)(dst + 4), h2); BBX_OBS_ST(dst + 6, x[6]); break;
    default: BBX_OBS_ST((bbx_obs_i4*)dst, lo); BBX_OBS_ST((bbx_obs_i4*)(dst + 4), hi); break;   // 8 variables
  }
}

// ------------------------------------------------------------------ environment view
// Kernel arguments are re-read from the kernarg segment where they are used, through a pointer the optimiser cannot see through
// (constant address space + uniform address = s_load from the scalar cache): otherwise every field of the by-value struct is
// loaded at kernel entry and stays live — in scalar registers, of which the step loop has none to spare (~400 of them spilled
// to vector lanes) — across the whole loop.  The kernels take the BbxParams struct as their FIRST argument: offset 0.
__device__ __forceinline__ const BbxParams& bbx_kparams() {
  const __attribute__((address_space(4))) BbxParams* q = (const __attribute__((address_space(4))) BbxParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return *(const BbxParams*)q;
}
template <int W> struct Env {
  static constexpr bool kCached = false;               // (see BEnvC in bbx_binom.h)
  BbxHdr* hdr;
  Mono<W>*lm, *slm, *lcm, *am, *hm;
  uint32_t *poff, *pairs;
  uint16_t *sidx, *plen, *psug, *pinv, *ac, *hc;
  uint8_t* cp;
};
template <int W> __device__ __forceinline__ Env<W> env_view(char* rec, const BbxLayout& L) {
  Env<W> e;
  e.hdr = (BbxHdr*)rec;
  e.lm = (Mono<W>*)(rec + L.off_lm); e.slm = (Mono<W>*)(rec + L.off_slm); e.lcm = (Mono<W>*)(rec + L.off_lcm);
  e.am = (Mono<W>*)(rec + L.off_am); e.hm = (Mono<W>*)(rec + L.off_hm);
  e.poff = (uint32_t*)(rec + L.off_poff); e.pairs = (uint32_t*)(rec + L.off_pairs);
  e.sidx = (uint16_t*)(rec + L.off_sidx); e.plen = (uint16_t*)(rec + L.off_plen); e.psug = (uint16_t*)(rec + L.off_psug);
  e.pinv = (uint16_t*)(rec + L.off_pinv); e.ac = (uint16_t*)(rec + L.off_ac); e.hc = (uint16_t*)(rec + L.off_hc);
  e.cp = (uint8_t*)(rec + L.off_cp);
  return e;
}

// a polynomial seen through a term multiplier:  element t = (c[t]*scale mod p) * x^(m[t]+shift)
template <int W> struct PView {
  const Mono<W>* m; const uint16_t* c; int n; Mono<W> shift; uint32_t scale;
  __device__ __forceinline__ Mono<W> mono(int t) const { return m_mul(m[t], shift); }
  __device__ __forceinline__ uint32_t coef(int t) const { return mulmod(c[t], scale); }
};

// out = A + B (both descending grevlex), equal monomials summed, zero sums dropped:
// Polynomial operator+ (polynomials.cpp:148-177) as a rank-and-compact wave merge.
// tm/tc: staging of at least A.n + B.n terms.  Returns the number of output terms, or -1 if it
// would exceed ocap.
template <int W>
__device__ int wave_merge(const PView<W>& A, const PView<W>& B, Mono<W>* tm, uint16_t* tc,
                          Mono<W>* om, uint16_t* oc, int ocap, unsigned long long* prof = nullptr) {
  const int lane = lane_id();
  unsigned long long t0_ = prof ? __builtin_amdgcn_s_memtime() : 0;
  const int na = A.n, nb = B.n, total = na + nb;
  // pass 1: every term finds its slot in the virtual merged sequence (short polynomials only: long ones take
  // wave_merge_tiled below)
  for (int x = lane; x < na; x += WAVE) {
    Mono<W> a = A.mono(x);
    int lo = 0, hi = nb;                       // #B terms strictly greater than a
    while (lo < hi) { int mid = (lo + hi) >> 1; if (m_gt(B.mono(mid), a)) lo = mid + 1; else hi = mid; }
    uint32_t c = A.coef(x);
    if (lo < nb && m_eq(B.mono(lo), a)) c = addmod(c, B.coef(lo));
    tm[x + lo] = a; tc[x + lo] = (uint16_t)c;  // c == 0 marks a hole
  }
  for (int y = lane; y < nb; y += WAVE) {
    Mono<W> b = B.mono(y);
    int lo = 0, hi = na;                       // #A terms >= b
    while (lo < hi) { int mid = (lo + hi) >> 1; if (m_gt(b, A.mono(mid))) hi = mid; else lo = mid + 1; }
    bool dup = lo > 0 && m_eq(A.mono(lo - 1), b);
    tm[y + lo] = b; tc[y + lo] = dup ? (uint16_t)0 : (uint16_t)B.coef(y);
  }
  wave_sync();
  if (prof) { unsigned long long t1_ = __builtin_amdgcn_s_memtime(); prof[0] += t1_ - t0_; t0_ = t1_; }
  // pass 2: squeeze the holes out
  int nout = 0;
  for (int base = 0; base < total; base += WAVE) {
    int idx = base + lane;
    uint32_t c = idx < total ? tc[idx] : 0u;
    uint64_t mask = ballot64(c != 0);
    if (c != 0) {
      int o = nout + prefix_of(mask, lane);
      if (o < ocap) { om[o] = tm[idx]; oc[o] = (uint16_t)c; }
    }
    nout += __popcll(mask);
  }
  wave_sync();
  if (prof) { unsigned long long t1_ = __builtin_amdgcn_s_memtime(); prof[1] += t1_ - t0_; }
  return nout > ocap ? -1 : nout;
}

// ------------------------------------------------------------------ long polynomials: merge-path tiles through LDS
// The rank-by-binary-search merge above gathers 16-byte terms at random from HBM/L2, one cache line per probe; for the
// long polynomials of cyclic-n (hundreds to thousands of terms) that line traffic is the whole cost (profiles/).
// Here the virtual merged sequence is cut into tiles of MT positions along merge-path diagonals (all tile boundaries
// are searched at once, one per lane), each tile's two contiguous input ranges are streamed into LDS with coalesced
// loads, ranks are computed against LDS, holes (cancelled or combined terms) are squeezed out with ballots and the
// tile is appended to the output — no staging pass through memory.
constexpr int MT = 184;                                   // virtual positions per tile: MT + 8 = 3 x 64, i.e. exactly three
                                                          // terms per lane (256 left a mostly empty fifth round per lane)
template <int W> __host__ __device__ constexpr int merge_lds_bytes() { return 2 * (MT + 8) * (4 * W + 2); }

// merge-path partition: lane l returns (i, j), i + j = d = (t0 + l) * MT (clamped), such that A[0..i) and B[0..j) are
// exactly the first d terms of the merge (ties: the A term first); an equal pair is never split across the boundary
template <int W>
__device__ __forceinline__ void merge_partition(const PView<W>& A, const PView<W>& B, int t0, int& bi, int& bj) {
  const int na = A.n, nb = B.n, total = na + nb;
  int d = (t0 + lane_id()) * MT; d = d < total ? d : total;
  int lo = d - nb > 0 ? d - nb : 0, hi = d < na ? d : na;
  while (__any(lo < hi)) {
    const bool act = lo < hi;
    const int mid = (lo + hi) >> 1;
    const Mono<W> am = A.mono(act ? mid : 0), bm = B.mono(act ? d - 1 - mid : 0);   // na, nb >= 1
    if (act) { if (!m_gt(bm, am)) lo = mid + 1; else hi = mid; }
  }
  bi = lo; bj = d - lo;
  const bool chk = bi > 0 && bj < nb;
  const Mono<W> am = A.mono(chk ? bi - 1 : 0), bm = B.mono(chk ? bj : 0);
  if (chk && m_eq(am, bm)) bj++;
}

// the same partition for up to 8 boundaries at once, 8 lanes per boundary probing 8 split points per step: the search
// range shrinks 8x per dependent memory round trip instead of 2x (these probes are the serial part of a merge)
template <int W>
__device__ __forceinline__ void merge_partition8(const PView<W>& A, const PView<W>& B, int t0, int& bi, int& bj) {
  const int na = A.n, nb = B.n, total = na + nb;
  const int lane = lane_id(), g = lane >> 3, s = lane & 7;
  int d = (t0 + g) * MT; d = d < total ? d : total;
  int lo = d - nb > 0 ? d - nb : 0, hi = d < na ? d : na;
  while (__any(lo < hi)) {
    const bool act = lo < hi;
    const int span = hi - lo;
    const int mid = lo + ((s * span) >> 3);                 // 8 probes across [lo, hi)
    const Mono<W> am = A.mono(act ? mid : 0), bm = B.mono(act ? d - 1 - mid : 0);
    const bool pred = act && !m_gt(bm, am);                 // A[mid] >= B[d-1-mid]: the boundary lies beyond mid
    const uint32_t bits = (uint32_t)(ballot64(pred) >> (g * 8)) & 0xffu;
    const int c = __popc(bits);                             // pred is monotone in mid: the true probes are a prefix
    if (act) {
      const int nlo = c > 0 ? lo + (((c - 1) * span) >> 3) + 1 : lo;
      const int nhi = c < 8 ? lo + ((c * span) >> 3) : hi;
      lo = nlo; hi = nhi;
    }
  }
  bi = lo; bj = d - lo;
  const bool chk = bi > 0 && bj < nb;
  const Mono<W> am = A.mono(chk ? bi - 1 : 0), bm = B.mono(chk ? bj : 0);
  if (chk && m_eq(am, bm)) bj++;
}

// one tile: ranges A[i0..i1) and B[j0..j1) are streamed into LDS (coalesced), ranked against each other there, merged
// with equal monomials combined and zero sums dropped, and written densely to (dm, dc).  Returns the number written
// (<= MT); nothing beyond `cap` terms is stored.
template <int W>
__device__ int merge_tile(const PView<W>& A, const PView<W>& B, int i0, int j0, int i1, int j1, char* lds,
                          Mono<W>* dm, uint16_t* dc, int cap) {
  const int lane = lane_id();
  Mono<W>* Lm = (Mono<W>*)lds;                             // tile terms: A range then B range
  Mono<W>* Tm = Lm + (MT + 8);                             // the tile in merged order, with holes
  uint16_t* Lc = (uint16_t*)(Tm + (MT + 8));
  uint16_t* Tc = Lc + (MT + 8);
  const int nat = i1 - i0, nbt = j1 - j0, nt = nat + nbt;  // nt <= MT + 1
  constexpr int MU = (MT + 8 + WAVE - 1) / WAVE;           // terms per lane
  Mono<W> mine[MU]; uint32_t mc[MU]; int lo2[MU];
#pragma unroll
  for (int u = 0; u < MU; u++) {                           // all loads issued, then consumed
    const int q = lane + u * WAVE;
    const bool isa = q < nat;
    const int src = q < nt ? (isa ? i0 + q : j0 + q - nat) : 0;
    // one load per lane from whichever polynomial the position belongs to (not one from each with a dummy index)
    const Mono<W>* pm = isa ? A.m + src : B.m + src;
    const uint16_t* pc = isa ? A.c + src : B.c + src;
    const Mono<W> mx = *pm;
    const uint32_t cx = *pc;
    Mono<W> sh; uint32_t sc = isa ? A.scale : B.scale;
#pragma unroll
    for (int i = 0; i < W; i++) sh.w[i] = isa ? A.shift.w[i] : B.shift.w[i];
    mine[u] = m_mul(mx, sh);
    mc[u] = mulmod(cx, sc);
    lo2[u] = 0;
  }
#pragma unroll
  for (int u = 0; u < MU; u++) { const int q = lane + u * WAVE; if (q < nt) { Lm[q] = mine[u]; Lc[q] = (uint16_t)mc[u]; } }
  wave_sync();
  int top = 1; { const int mx = nat > nbt ? nat : nbt; while (top * 2 <= mx) top *= 2; }
  for (int step = top; step > 0; step >>= 1) {             // rank of every term in the OTHER range, lockstep
#pragma unroll
    for (int u = 0; u < MU; u++) {
      const int q = lane + u * WAVE;
      const bool isa = q < nat;
      const int lim = isa ? nbt : nat, off = isa ? nat : 0;
      const int idx = lo2[u] + step;
      const bool ok = q < nt && idx <= lim;
      const Mono<W> pm = Lm[ok ? off + idx - 1 : 0];
      // A term: count B terms strictly greater;  B term: count A terms greater or equal
      const bool adv = isa ? m_gt(pm, mine[u]) : !m_gt(mine[u], pm);
      lo2[u] = (ok && adv) ? idx : lo2[u];
    }
  }
#pragma unroll
  for (int u = 0; u < MU; u++) {
    const int q = lane + u * WAVE;
    if (q < nt) {
      uint32_t c = mc[u];
      int pos;
      if (q < nat) {
        pos = q + lo2[u];
        if (lo2[u] < nbt && m_eq(Lm[nat + lo2[u]], mine[u])) c = addmod(c, Lc[nat + lo2[u]]);
      } else {
        pos = (q - nat) + lo2[u];
        if (lo2[u] > 0 && m_eq(Lm[lo2[u] - 1], mine[u])) c = 0;   // combined into the equal A term
      }
      Tm[pos] = mine[u]; Tc[pos] = (uint16_t)c;                    // c == 0 marks a hole
    }
  }
  wave_sync();
  int cnt = 0;
  for (int base = 0; base < nt; base += WAVE) {            // squeeze the holes out
    const int idx = base + lane;
    const uint32_t c = idx < nt ? Tc[idx] : 0u;
    const uint64_t mask = ballot64(c != 0);
    if (c != 0) {
      const int o = cnt + prefix_of(mask, lane);
      if (o < cap) { dm[o] = Tm[idx]; dc[o] = (uint16_t)c; }
    }
    cnt += __popcll(mask);
  }
  wave_sync();
  return cnt;
}

// single-wave driver: tiles in order, appended directly to the output
template <int W>
__device__ int wave_merge_tiled(const PView<W>& A, const PView<W>& B, char* lds, Mono<W>* om, uint16_t* oc, int ocap) {
  const int total = A.n + B.n;
  const int ntiles = (total + MT - 1) / MT;
  int nout = 0;
  for (int t0 = 0; t0 < ntiles; t0 += 63) {               // 64 boundaries -> 63 tiles per batch
    int bi, bj;
    merge_partition<W>(A, B, t0, bi, bj);
    const int batch = ntiles - t0 < 63 ? ntiles - t0 : 63;
    for (int tt = 0; tt < batch; tt++) {
      const int i0 = __builtin_amdgcn_readlane(bi, tt), j0 = __builtin_amdgcn_readlane(bj, tt);
      const int i1 = __builtin_amdgcn_readlane(bi, tt + 1), j1 = __builtin_amdgcn_readlane(bj, tt + 1);
      const int room = ocap - nout > 0 ? ocap - nout : 0;
      nout += merge_tile<W>(A, B, i0, j0, i1, j1, lds, om + (nout < ocap ? nout : 0), oc + (nout < ocap ? nout : 0), room);
    }
  }
  return nout > ocap ? -1 : nout;
}

// ------------------------------------------------------------------ ideal generation on the device
// The reference draws every new ideal from std::default_random_engine (minstd_rand0) through libstdc++ 11's
// uniform_int_distribution / discrete_distribution (generate_canonical<double, 53>: two engine draws) — restated here
// from the same published algorithms as the host generators (bbx_ideals.cpp), operation for operation, so that a
// seeded environment sees the same ideals wherever they are drawn.  Everything is wave-uniform: scalar registers, scalar
// loads from the immutable table.  Doubles: plain IEEE operations, no contraction.
__device__ __forceinline__ uint32_t gen_next(uint32_t& x) {                 // x <- 16807 x mod (2^31 - 1)
  const uint64_t pr = (uint64_t)x * 16807u;
  uint32_t s = (uint32_t)(pr & 0x7fffffffu) + (uint32_t)(pr >> 31);        // 2^31 = 1 (mod 2^31 - 1)
  if (s >= 2147483647u) s -= 2147483647u;
  x = s;
  return s;
}
// ret / scaling for ret < 2^31 with magic = floor(2^32 / scaling): the estimate is at most one too small
__device__ __forceinline__ uint32_t gen_div(uint32_t ret, uint32_t scaling, uint32_t magic) {
  uint32_t q = (uint32_t)(((uint64_t)ret * magic) >> 32);
  if (ret - q * scaling >= scaling) q++;
  return q;
}
__device__ __forceinline__ uint32_t gen_uniform(uint32_t& x, uint32_t scaling, uint32_t past, uint32_t magic) {   // uniform_int_dist.h, downscaling
  uint32_t ret;
  do ret = gen_next(x) - 1u; while (ret >= past);
  return gen_div(ret, scaling, magic);
}
__device__ __forceinline__ double gen_canonical(uint32_t& x) {             // random.tcc generate_canonical, k = 2
#pragma clang fp contract(off)
  const double R = 2147483646.0;
  double sum = (double)(gen_next(x) - 1u);
  sum = sum + (double)(gen_next(x) - 1u) * R;
  double ret = sum / (R * R);
  if (ret >= 1.0) ret = 0x1.fffffffffffffp-1;                               // nextafter(1, 0)
  return ret;
}
// The small tables live one entry per lane for the duration of a reset (two coalesced loads): the cumulative
// probabilities, so that std::lower_bound is a compare + ballot + popcount, and the per-degree rows, fetched with
// v_readlane.  Only the chosen monomials themselves are loaded (scalar loads) while drawing.
struct GenLanes { double cp; uint32_t off, scaling, past, magic; };
__device__ __forceinline__ GenLanes gen_lanes(const uint32_t* g) {
  const int lane = lane_id();
  GenLanes r;
  r.cp = *(const double*)(g + BBX_GEN_CP + 2 * lane);                       // +inf beyond the last entry
  const uint4 row = *(const uint4*)(g + BBX_GEN_DEG + 8 * lane);
  r.off = row.x; r.scaling = row.z; r.past = row.w; r.magic = g[BBX_GEN_DEG + 8 * lane + 4];
  return r;
}
__device__ __forceinline__ int gen_degree(uint32_t& x, const GenLanes& L, int ncp) {   // discrete_distribution::operator()
  if (ncp == 0) return 0;
  const double pr = gen_canonical(x);
  return __popcll(ballot64(L.cp < pr));                                     // std::lower_bound(cp, cp + ncp, pr) - cp
}
template <int W>
__device__ __forceinline__ Mono<W> gen_choice(uint32_t& x, const uint32_t* g, const GenLanes& L, int d) {   // choice(bases[d], rng), ideals.h:68-73
  const uint32_t scaling = (uint32_t)__builtin_amdgcn_readlane((int)L.scaling, d), past = (uint32_t)__builtin_amdgcn_readlane((int)L.past, d);
  const uint32_t magic = (uint32_t)__builtin_amdgcn_readlane((int)L.magic, d), off = (uint32_t)__builtin_amdgcn_readlane((int)L.off, d);
  const uint32_t j = off + gen_uniform(x, scaling, past, magic);
  Mono<W> m;
#pragma unroll
  for (int i = 0; i < W; i++) m.w[i] = ldc(g + BBX_GEN_MONO + (size_t)j * W + i);
  return m;
}
// one generator of the ideal: {1 * bigger monomial, c * smaller monomial} (ideals.cpp:168-201); false after 1000 trials
template <int W>
__device__ __forceinline__ bool gen_binomial(uint32_t& x, const uint32_t* g, const GenLanes& L, uint32_t flags, int ncp,
                                             Mono<W>& lead, Mono<W>& tail, uint32_t& c) {
  c = (flags & 2u) ? BBX_P - 1u : 1u + gen_uniform(x, 67104u, 2147462208u, 64004u);   // uniform_int_distribution(1, P - 1); 2^32 / 67104 = 64004
  int d1, d2;
  if (flags & 1u) d1 = d2 = gen_degree(x, L, ncp);
  else { d1 = gen_degree(x, L, ncp); d2 = gen_degree(x, L, ncp); }
  for (int trials = 0; trials < 1000; trials++) {
    const Mono<W> m1 = gen_choice<W>(x, g, L, d1), m2 = gen_choice<W>(x, g, L, d2);
    if (m_gt(m2, m1)) { lead = m2; tail = m1; return true; }
    if (m_gt(m1, m2)) { lead = m1; tail = m2; return true; }
  }
  return false;
}

// sort_input (buchberger.cpp:299-303: std::sort of the drawn generators by lead monomial, ascending).  For at most 16
// elements libstdc++'s std::sort IS its insertion sort (introsort's threshold, bits/stl_algo.h:1880-1886), which never
// moves an element past an equal one: the order is the stable one.  Lane f holds LM(generator f); returns the position of
// the lane's generator in that order.  (More than 16 generators: the host generators sort, bbx_api.cpp.)
template <int W> __device__ __forceinline__ int gen_sorted_rank(const Mono<W>& mine, int npoly) {
  const int lane = lane_id();
  int rank = 0;
  for (int g = 0; g < npoly; g++) {
    Mono<W> o;
#pragma unroll
    for (int q = 0; q < W; q++) o.w[q] = (uint32_t)__builtin_amdgcn_readlane((int)mine.w[q], g);
    rank += (m_gt(mine, o) || (m_eq(mine, o) && g < lane)) ? 1 : 0;
  }
  return rank;
}

// poisson_distribution<int>(lambda)(rng) for lambda < 12 (random.tcc): multiply canonical draws until below exp(-lambda)
__device__ __forceinline__ int gen_poisson(uint32_t& x, double lm_thr) {
#pragma clang fp contract(off)
  int k = 0;
  double prod = 1.0;
  do { prod = prod * gen_canonical(x); k += 1; } while (prod > lm_thr);
  return k - 1;
}

// ------------------------------------------------------------------ std::sort of basis indices by lead monomial
// libstdc++ 11's std::sort (bits/stl_algo.h: introsort = median-of-three quicksort with a depth limit of 2 log2 n and a
// heapsort fallback, segments of <= 16 elements left to a final insertion sort), restated on an index array and run by ONE
// lane: where the reference sorts polynomials by lead monomial (buchberger.cpp:102-104 minimalize, :157-158 buchberger)
// the order of elements with EQUAL lead monomials is whatever that algorithm leaves, and results depend on it.
// less(a, b) = LM(G[a]) < LM(G[b]).
template <int W> struct SortCtx {
  const Mono<W>* lm; uint16_t* v;
  __device__ __forceinline__ bool less(int a, int b) const { return m_gt(lm[b], lm[a]); }
};
template <int W> __device__ void ss_unguarded_linear_insert(const SortCtx<W>& c, int last) {
  const int val = c.v[last];
  int next = last - 1;
  while (c.less(val, c.v[next])) { c.v[last] = c.v[next]; last = next; next--; }
  c.v[last] = (uint16_t)val;
}
template <int W> __device__ void ss_insertion_sort(const SortCtx<W>& c, int first, int last) {
  if (first == last) return;
  for (int i = first + 1; i != last; i++) {
    if (c.less(c.v[i], c.v[first])) {
      const int val = c.v[i];
      for (int j = i; j > first; j--) c.v[j] = c.v[j - 1];               // move_backward(first, i, i + 1)
      c.v[first] = (uint16_t)val;
    } else ss_unguarded_linear_insert<W>(c, i);
  }
}
template <int W> __device__ void ss_adjust_heap(const SortCtx<W>& c, int first, int hole, int len, int value) {
  const int top = hole;
  int child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (c.less(c.v[first + child], c.v[first + child - 1])) child--;
    c.v[first + hole] = c.v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    c.v[first + hole] = c.v[first + child - 1];
    hole = child - 1;
  }
  int parent = (hole - 1) / 2;                                            // __push_heap
  while (hole > top && c.less(c.v[first + parent], value)) {
    c.v[first + hole] = c.v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  c.v[first + hole] = (uint16_t)value;
}
template <int W> __device__ void ss_heapsort(const SortCtx<W>& c, int first, int last) {   // __partial_sort(first, last, last)
  const int len = last - first;
  if (len >= 2)
    for (int parent = (len - 2) / 2;; parent--) { ss_adjust_heap<W>(c, first, parent, len, c.v[first + parent]); if (parent == 0) break; }
  while (last - first > 1) {
    --last;
    const int value = c.v[last];
    c.v[last] = c.v[first];
    ss_adjust_heap<W>(c, first, 0, last - first, value);
  }
}
template <int W> __device__ void ss_std_sort(const SortCtx<W>& c, int n) {
  if (n <= 0) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  // __introsort_loop: recursion on the right part, iteration on the left — the segments are independent, so an explicit
  // stack serves (at most one entry per level: the depth limit bounds it)
  int sf[48], sl[48], sd[48], sp = 0;
  sf[0] = 0; sl[0] = n; sd[0] = 2 * lg; sp = 1;
  while (sp > 0) {
    sp--;
    int first = sf[sp], last = sl[sp], depth = sd[sp];
    while (last - first > 16) {
      if (depth == 0) { ss_heapsort<W>(c, first, last); break; }
      --depth;
      const int mid = first + (last - first) / 2, a = first + 1, b = mid, cc = last - 1;
      auto swp = [&](int x, int y) { const uint16_t t = c.v[x]; c.v[x] = c.v[y]; c.v[y] = t; };
      if (c.less(c.v[a], c.v[b])) {                                       // __move_median_to_first(first, first + 1, mid, last - 1)
        if (c.less(c.v[b], c.v[cc])) swp(first, b); else if (c.less(c.v[a], c.v[cc])) swp(first, cc); else swp(first, a);
      } else if (c.less(c.v[a], c.v[cc])) swp(first, a);
      else if (c.less(c.v[b], c.v[cc])) swp(first, cc);
      else swp(first, b);
      int lo = first + 1, hi = last;                                      // __unguarded_partition(first + 1, last, pivot = first)
      for (;;) {
        while (c.less(c.v[lo], c.v[first])) lo++;
        --hi;
        while (c.less(c.v[first], c.v[hi])) --hi;
        if (!(lo < hi)) break;
        swp(lo, hi);
        lo++;
      }
      if (sp < 48) { sf[sp] = lo; sl[sp] = last; sd[sp] = depth; sp++; }
      last = lo;
    }
  }
  if (n > 16) {                                                            // __final_insertion_sort
    ss_insertion_sort<W>(c, 0, 16);
    for (int i = 16; i != n; i++) ss_unguarded_linear_insert<W>(c, i);
  } else ss_insertion_sort<W>(c, 0, n);
}
// ------------------------------------------------------------------ update()   buchberger.cpp:52-99
// Adds the polynomial whose lead monomial is lmf as G[m] (the caller has already stored its terms and
// metadata) and updates the pair set.  Returns false on capacity overflow.
#ifdef BBX_PROF_BUILD
#define USTAMP(slot) do { if (prof) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof[slot] += t_ - *plast; *plast = t_; } } while (0)
#else
#define USTAMP(slot) do {} while (0)
#endif
// peel_lds: per-wave LDS scratch of WAVE * UPD_CHUNKS monomials for the Gebauer-Moeller peel (or null: the record's
// scratch arrays in HBM are used)
constexpr int UPD_CHUNKS = 8;
template <int W> __host__ __device__ constexpr int update_lds_bytes() { return WAVE * UPD_CHUNKS * 4 * W; }
template <int W, class EnvT>
__device__ bool wave_update(EnvT& e, const BbxLayout& L, int& nG, int& nP, const Mono<W> lmf, int elim, int* status,
                            char* peel_lds = nullptr, unsigned long long* prof = nullptr, unsigned long long* plast = nullptr,
                            int* first_drop = nullptr) {
  // first_drop: out, the index of the first old pair that was dropped (the pair list is unchanged in front of it), or the
  // old |P| when none was — what an observation written incrementally needs to know
  const int lane = lane_id();
  const int m = nG;
  int fdrop = nP;
  if (elim == BBX_ELIM_GM) {
    // (70-76) drop old pairs (i,j) with LM f | lcm_ij, lcm_ij != lcm_if, lcm_ij != lcm_jf  — stable
    // Four chunks of 64 pairs per trip: all pair words are loaded together, then all lead-monomial gathers are in flight
    // together (two dependent trips to memory per 256 pairs instead of two per 64); the survivors are then written, in
    // order, behind the write cursor — which never passes a pair that has not been read yet.
    int w = 0;
    constexpr int UF = 4;
    for (int base = 0; base < nP; base += WAVE * UF) {
      uint32_t pr[UF]; Mono<W> li[UF], lj[UF]; bool in[UF];
#pragma unroll
      for (int u = 0; u < UF; u++) { const int k = base + u * WAVE + lane; in[u] = k < nP; pr[u] = in[u] ? e.pairs[k] : 0u; }
      if constexpr (EnvT::kCached) e.lm.gather_pairs(pr, li, lj);
      else {
#pragma unroll
        for (int u = 0; u < UF; u++) { li[u] = e.lm[pr[u] & 0xffffu]; lj[u] = e.lm[pr[u] >> 16]; }   // (index 0 for absent lanes)
      }
      wave_sync();
#pragma unroll
      for (int u = 0; u < UF; u++) {
        if (base + u * WAVE < nP) {
          const Mono<W> l = m_lcm(li[u], lj[u]);
          const bool drop = m_divides(lmf, l) && !m_eq(l, m_lcm(li[u], lmf)) && !m_eq(l, m_lcm(lj[u], lmf));
          const bool keep = in[u] && !drop;
          const uint64_t mask = ballot64(keep);
          if (first_drop) {
            const uint64_t dmask = ballot64(in[u] && drop);
            if (dmask && fdrop > base + u * WAVE) { const int fd = base + u * WAVE + __builtin_ctzll(dmask); fdrop = fd < fdrop ? fd : fdrop; }
          }
          if (keep) e.pairs[w + prefix_of(mask, lane)] = pr[u];
          w = uni(w + __popcll(mask));   // pinned: the optimiser otherwise threads the count through the per-lane branch
                                         // above and the uniformity analysis gives up on |P| (-> exec-masked code)
        }
      }
      wave_sync();
    }
    nP = w;
    if (first_drop) *first_drop = fdrop;
    USTAMP(8);
    if ((EnvT::kCached && m <= 32 * WAVE) || (!EnvT::kCached && peel_lds != nullptr && m <= WAVE * UPD_CHUNKS)) {
      // (78-91) new pairs (i, m), everything on chip: L_i = lcm(LM G[i], LM f) in the wave's LDS scratch (element i at slot
      // i: conflict-free 16-byte reads), the candidate / coprime / emit flags one bit per element in three registers
      // (lane l owns elements l, l + 64, ...), a bucket's lcm travels by v_readlane.  The std::map walk keeps exactly the
      // lcms minimal under divisibility among the distinct values; they are peeled by increasing degree (see below).
      // Environments with an LDS copy of their lead monomials (EnvT::kCached) need no scratch: L_i is formed from the copy
      // wherever it is read.
      typedef uint32_t bbx_u32xW __attribute__((ext_vector_type(W)));
      __attribute__((address_space(3))) bbx_u32xW* Ll = (__attribute__((address_space(3))) bbx_u32xW*)peel_lds;
      const int nch = (m + WAVE - 1) / WAVE;
      uint32_t candb = 0, cpb = 0, emitb = 0;
      for (int u = 0; u < nch; u++) {
        const int i = u * WAVE + lane;
        const bool v = i < m;
        const Mono<W> li = v ? (Mono<W>)e.lm[i] : m_zero<W>();
        if constexpr (!EnvT::kCached) {
          const Mono<W> Li = m_lcm(li, lmf);
          bbx_u32xW pk;
#pragma unroll
          for (int q = 0; q < W; q++) pk[q] = Li.w[q];
          Ll[i] = pk;
        }
        candb |= v ? (1u << u) : 0u;
        cpb |= (v && m_coprime(li, lmf)) ? (1u << u) : 0u;
      }
      wave_sync();
      USTAMP(9);
      auto ldsL = [&](int i) {
        if constexpr (EnvT::kCached) return m_lcm((Mono<W>)e.lm[i < m ? i : 0], lmf);
        else { const bbx_u32xW pk = Ll[i]; Mono<W> r; for (int q = 0; q < W; q++) r.w[q] = pk[q]; return r; }
      };
      for (;;) {
        uint32_t dm = 0xFFFFFFFFu;
        for (int u = 0; u < nch; u++) { const uint32_t d = m_deg(ldsL(u * WAVE + lane)); dm = ((candb >> u) & 1u) && d < dm ? d : dm; }
        const uint32_t dmin = wave_min32(dm);
        if (dmin == 0xFFFFFFFFu) break;
        for (int u = 0; u < nch; u++) {
          const Mono<W> Lu = ldsL(u * WAVE + lane);
          uint64_t surv = ballot64(((candb >> u) & 1u) && m_deg(Lu) == dmin);
          while (surv) {
            const int sl = __builtin_ctzll(surv);
            Mono<W> Ls;
#pragma unroll
            for (int q = 0; q < W; q++) Ls.w[q] = (uint32_t)__builtin_amdgcn_readlane((int)Lu.w[q], sl);
            uint64_t any_cp = 0;
            for (int v = 0; v < nch; v++) {
              const Mono<W> Lv = ldsL(v * WAVE + lane);
              const bool eq = m_eq(Lv, Ls);
              if (m_divides(Ls, Lv)) candb &= ~(1u << v);
              any_cp |= ballot64(eq && ((cpb >> v) & 1u));
              if (v == u) surv &= ~ballot64(eq);
            }
            if (any_cp == 0 && lane == sl) emitb |= 1u << u;
          }
        }
      }
      USTAMP(10);
      for (int u = 0; u < nch; u++) {
        const bool emit = (emitb >> u) & 1u;
        const uint64_t mask = ballot64(emit);
        const int cnt = __popcll(mask);
        if (nP + cnt > (int)L.maxP) { *status = BBX_ST_P_FULL; return false; }
        if (emit) e.pairs[nP + prefix_of(mask, lane)] = (uint32_t)(u * WAVE + lane) | ((uint32_t)m << 16);  // (92) ascending i
        nP = uni(nP + cnt);
      }
      wave_sync();
      USTAMP(11);
      return true;
    }
    // (78-81) lcm_i = lcm(LM G[i], LM f); flags: 1 = G[i] coprime to f, 2 = still a candidate, 4 = emits a pair
    for (int i = lane; i < m; i += WAVE) {
      Mono<W> li = e.lm[i];
      e.lcm[i] = m_lcm(li, lmf);
      e.cp[i] = (uint8_t)((m_coprime(li, lmf) ? 1 : 0) | 2);
    }
    wave_sync();
    USTAMP(9);
    // (82-91) The std::map walk keeps exactly the lcms that are minimal under divisibility among the distinct
    // values (a proper divisor has smaller degree, hence comes earlier in grevlex).  They are peeled by increasing
    // degree: every candidate of minimal degree is minimal; its bucket of equal lcms emits (smallest index, m)
    // unless a member is coprime to f (88-89), and all multiples of it stop being candidates.  Every lane only
    // ever reads and writes the flags of its own indices (i = lane mod 64), the bucket's lcm travels by readlane.
    for (;;) {
      uint32_t dm = 0xFFFFFFFFu;
      for (int i = lane; i < m; i += WAVE) if (e.cp[i] & 2) { uint32_t d = m_deg(e.lcm[i]); dm = d < dm ? d : dm; }
      const uint32_t dmin = wave_min32(dm);
      if (dmin == 0xFFFFFFFFu) break;
      for (int base = 0; base < m; base += WAVE) {
        const int i = base + lane;
        Mono<W> Li = m_zero<W>();
        bool sv = false;
        if (i < m) { Li = e.lcm[i]; sv = (e.cp[i] & 2) && m_deg(Li) == dmin; }
        uint64_t surv = ballot64(sv);
        while (surv) {
          const int sl = __builtin_ctzll(surv);
          Mono<W> Ls;
#pragma unroll
          for (int q = 0; q < W; q++) Ls.w[q] = (uint32_t)__builtin_amdgcn_readlane((int)Li.w[q], sl);
          bool any_cp = false;
          for (int b2 = 0; b2 < m; b2 += WAVE) {
            const int j = b2 + lane;
            bool eq = false;
            if (j < m) {
              const Mono<W> Lj = e.lcm[j];
              const uint8_t fj = e.cp[j];
              eq = m_eq(Lj, Ls);
              if (m_divides(Ls, Lj)) e.cp[j] = (uint8_t)(fj & ~2);
              any_cp |= eq && (fj & 1);
            }
            if (b2 == base) surv &= ~ballot64(eq);
          }
          if (ballot64(any_cp) == 0 && lane == sl) e.cp[i] |= 4;
        }
      }
    }
    USTAMP(10);
    for (int base = 0; base < m; base += WAVE) {
      const int i = base + lane;
      const bool emit = i < m && (e.cp[i] & 4);
      const uint64_t mask = ballot64(emit);
      const int cnt = __popcll(mask);
      if (nP + cnt > (int)L.maxP) { *status = BBX_ST_P_FULL; return false; }
      if (emit) e.pairs[nP + prefix_of(mask, lane)] = (uint32_t)i | ((uint32_t)m << 16);  // (92) ascending i
      nP = uni(nP + cnt);
    }
  } else {
    for (int base = 0; base < m; base += WAVE) {
      int i = base + lane;
      bool emit = false;
      if (i < m) emit = (elim == BBX_ELIM_NONE) ? true : !m_coprime(e.lm[i], lmf);  // 58-68
      uint64_t mask = ballot64(emit);
      int cnt = __popcll(mask);
      if (nP + cnt > (int)L.maxP) { *status = BBX_ST_P_FULL; return false; }
      if (emit) e.pairs[nP + prefix_of(mask, lane)] = (uint32_t)i | ((uint32_t)m << 16);
      nP = uni(nP + cnt);
    }
  }
  wave_sync();
  USTAMP(11);
  return true;
}

// insert G[g] into the reducer order: std::upper_bound by lead monomial, buchberger.cpp:309-311, 323-326
template <int W>
__device__ void wave_insert_reducer(Env<W>& e, int nR, int g, const Mono<W> lmf, int sort_reducers) {
  const int lane = lane_id();
  int pos = nR;
  if (sort_reducers) {
    pos = 0;                                   // #reducers with LM <= lmf
    for (int base = 0; base < nR; base += WAVE) {
      int k = base + lane;
      bool le = k < nR && !m_gt(e.slm[k], lmf);
      pos += __popcll(ballot64(le));
    }
    for (int hi = nR; hi > pos; hi -= WAVE) { // shift [pos, nR) up by one, top chunk first
      int k = hi - 1 - lane;
      Mono<W> v; uint16_t s = 0;
      if (k >= pos) { v = e.slm[k]; s = e.sidx[k]; }
      wave_sync();
      if (k >= pos) { e.slm[k + 1] = v; e.sidx[k + 1] = s; }
      wave_sync();
    }
  }
  if (lane == 0) { e.slm[pos] = lmf; e.sidx[pos] = (uint16_t)g; }
  wave_sync();
}

// append a polynomial (terms at sm/sc) to the basis: arena copy, metadata, update(), reducer insert
template <int W>
__device__ bool wave_add_poly(Env<W>& e, const BbxLayout& L, int& nG, int& nP, int& arena_used,
                              const Mono<W>* sm, const uint16_t* sc, int n, int sugar,
                              int elim, int sort_reducers, int* status, bool in_place = false) {
  const int lane = lane_id();
  if (nG >= (int)L.maxG) { *status = BBX_ST_G_FULL; return false; }
  if (n > 65535) { *status = BBX_ST_POLY_LIMIT; return false; }          // plen[] is 16 bits
  if (arena_used + n > (int)L.arena) { *status = BBX_ST_ARENA_FULL; return false; }
  const int g = nG, off = arena_used;
  // in_place: the terms already sit at the arena's end (the wide class builds the remainder there)
  if (!in_place) for (int t = lane; t < n; t += WAVE) { e.am[off + t] = sm[t]; e.ac[off + t] = sc[t]; }
  const Mono<W> lmf = sm[0];
  const uint32_t lc = sc[0];
  if (lane == 0) {
    e.lm[g] = lmf; e.poff[g] = (uint32_t)off; e.plen[g] = (uint16_t)n; e.psug[g] = (uint16_t)sugar;
    e.pinv[g] = (uint16_t)invmod(lc);
  }
  wave_sync();
  if (!wave_update<W>(e, L, nG, nP, lmf, elim, status)) return false;
  wave_insert_reducer<W>(e, g, g, lmf, sort_reducers);
  nG = g + 1;
  arena_used = off + n;
  return true;
}

// BuchbergerEnv::reset (buchberger.cpp:299-315) from the next host-generated ideal(s) of the queue.
// Returns false if the queue ran dry (status STARVED) or on overflow.
// One generator of a random ideal, drawn on the device, as a sorted polynomial in the scratch arrays (sm, sc):
//  * binomial distributions: {1 * bigger monomial, c * smaller} (gen_binomial);
//  * polynomial distributions (ideals.cpp:203-231): 2 + Poisson terms c * monomial summed with Polynomial operator+, then
//    scaled by 1/LC.  The sum over single terms is order-independent (coefficients add mod P per distinct monomial, a
//    monomial whose total is 0 disappears), so it is formed in one go: lane j holds term j, every lane adds up the
//    coefficients of the terms equal to its own, the first lane of each class with a non-zero total keeps it, and the
//    rank among the kept ones (count of greater monomials) is the position.  sugar = the largest degree DRAWN, cancelled
//    terms included (operator+ takes the max of the operands' sugars).
// Returns the number of terms (0: failure, *status set).
template <int W>
__device__ int gen_polynomial(uint32_t& x, const uint32_t* g, const GenLanes& GL, uint32_t flags, int ncp, double lm_thr,
                              Mono<W>* sm, uint16_t* sc, int maxT, int* sugar, int* status) {
  const int lane = lane_id();
  if (!(flags & 4u)) {
    Mono<W> lead, tail; uint32_t c;
    if (!gen_binomial<W>(x, g, GL, flags, ncp, lead, tail, c)) { *status = BBX_ST_GEN_FAIL; return 0; }
    if (lane == 0) { sm[0] = lead; sc[0] = 1; sm[1] = tail; sc[1] = (uint16_t)c; }
    *sugar = (int)m_deg(lead);
    wave_sync();
    return 2;
  }
  const int terms = 2 + gen_poisson(x, lm_thr);
  if (terms > WAVE || terms > maxT) { *status = BBX_ST_POLY_TOO_LONG; return 0; }
  int d = gen_degree(x, GL, ncp);
  Mono<W> mine = m_zero<W>(); uint32_t myc = 0;
  int maxdeg = 0;
  for (int j = 0; j < terms; j++) {
    const uint32_t c = 1u + gen_uniform(x, 67104u, 2147462208u, 64004u);
    const Mono<W> m = gen_choice<W>(x, g, GL, d);
    if (lane == j) { mine = m; myc = c; }
    maxdeg = d > maxdeg ? d : maxdeg;
    if (!(flags & 1u)) d = gen_degree(x, GL, ncp);
  }
  uint32_t total = 0; int first = lane;
  for (int k = 0; k < terms; k++) {
    Mono<W> mk;
#pragma unroll
    for (int i = 0; i < W; i++) mk.w[i] = (uint32_t)__builtin_amdgcn_readlane((int)mine.w[i], k);
    const uint32_t ck = (uint32_t)__builtin_amdgcn_readlane((int)myc, k);
    if (m_eq(mk, mine)) { total = addmod(total, ck); first = k < first ? k : first; }
  }
  const bool keep = lane < terms && first == lane && total != 0;
  const uint64_t km = ballot64(keep);
  int rank = 0;
  for (int k = 0; k < terms; k++) {
    if (!((km >> k) & 1ull)) continue;
    Mono<W> mk;
#pragma unroll
    for (int i = 0; i < W; i++) mk.w[i] = (uint32_t)__builtin_amdgcn_readlane((int)mine.w[i], k);
    if (m_gt(mk, mine)) rank++;
  }
  const int n = __popcll(km);
  if (n == 0) { *status = BBX_ST_GEN_ZERO; return 0; }
  // 1 / LC: the leading term is the kept one of rank 0
  const uint64_t lead_mask = ballot64(keep && rank == 0);
  const uint32_t lc = (uint32_t)__builtin_amdgcn_readlane((int)total, __builtin_ctzll(lead_mask));
  const uint32_t inv = invmod(lc);
  if (keep) { sm[rank] = mine; sc[rank] = (uint16_t)mulmod(total, inv); }
  *sugar = maxdeg;
  wave_sync();
  return n;
}

// BuchbergerEnv::reset (buchberger.cpp:299-315) from ideals drawn on the device (p.gen) or the next host-generated
// ideal(s) of the queue.  Returns false if the queue ran dry (status STARVED), the generator failed or on overflow.
template <int W>
__device__ bool wave_reset(Env<W>& e, const BbxParams& p, const BbxLayout& L, int env, int& nG, int& nP, int& arena_used, int& q_head, int* status,
                           uint32_t& gen_state) {
  const int lane = lane_id();
  if (p.gen) {
    const int npoly = (int)ldc(p.gen + 2), ncp = (int)ldc(p.gen + 4);
    const uint32_t gflags = ldc(p.gen + 3);
    const double lm_thr = __builtin_bit_cast(double, ((uint64_t)ldc(p.gen + 7) << 32) | ldc(p.gen + 6));
    const GenLanes GL = gen_lanes(p.gen);
    uint32_t x = (uint32_t)uni((int)gen_state);
    for (;;) {
      const uint32_t x_start = x;
      nG = 0; nP = 0; arena_used = 0;
      if (p.sort_input) {
        // all generators are drawn first, back to back in the scratch polynomials, then enter in sorted order
        Mono<W> lmine = m_zero<W>(); int nmine = 0, smine = 0, omine = 0, at = 0;
        for (int f = 0; f < npoly; f++) {
          int sugar = 0;
          if (at + ((gflags & 4u) ? WAVE : 2) > 5 * (int)L.maxT) { *status = BBX_ST_POLY_TOO_LONG; gen_state = x_start; return false; }
          const int n = gen_polynomial<W>(x, p.gen, GL, gflags, ncp, lm_thr, e.hm + at, e.hc + at, (int)L.maxT, &sugar, status);
          if (n == 0) { gen_state = x; return false; }
          const Mono<W> lead = e.hm[at];
          if (lane == f) { lmine = lead; nmine = n; smine = sugar; omine = at; }
          at += n;
        }
        const int rank = gen_sorted_rank<W>(lmine, npoly);
        for (int r = 0; r < npoly; r++) {
          const int src = __builtin_ctzll(ballot64(lane < npoly && rank == r));
          const int n = __builtin_amdgcn_readlane(nmine, src), sugar = __builtin_amdgcn_readlane(smine, src), off = __builtin_amdgcn_readlane(omine, src);
          if (!wave_add_poly<W>(e, L, nG, nP, arena_used, e.hm + off, e.hc + off, n, sugar, p.elim, p.sort_reducers, status)) { gen_state = x_start; return false; }
        }
      } else
      for (int f = 0; f < npoly; f++) {
        int sugar = 0;
        const int n = gen_polynomial<W>(x, p.gen, GL, gflags, ncp, lm_thr, e.hm, e.hc, (int)L.maxT, &sugar, status);
        if (n == 0) { gen_state = x; return false; }
        if (!wave_add_poly<W>(e, L, nG, nP, arena_used, e.hm, e.hc, n, sugar, p.elim, p.sort_reducers, status)) { gen_state = x_start; return false; }
      }
      if (nP != 0) { gen_state = x; return true; }   // 313-314: redraw while the pair set is empty
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
    nG = 0; nP = 0; arena_used = 0;
    const int npoly = (int)ldc(slot);
    const uint32_t* w = slot + 1;
    Mono<W>* sm = e.hm; uint16_t* sc = e.hc;   // stage each generator's terms in scratch
    for (int f = 0; f < npoly; f++) {
      const int n = (int)ldc(w), sugar = (int)ldc(w + 1);
      w += 2;
      if (n > (int)L.maxT) { *status = BBX_ST_POLY_TOO_LONG; return false; }
      for (int t = lane; t < n; t += WAVE) {
        const uint32_t* tw = w + (size_t)t * (1 + W);
        Mono<W> mm;
#pragma unroll
        for (int i = 0; i < W; i++) mm.w[i] = tw[1 + i];
        sm[t] = mm; sc[t] = (uint16_t)tw[0];
      }
      wave_sync();
      if (!wave_add_poly<W>(e, L, nG, nP, arena_used, sm, sc, n, sugar, p.elim, p.sort_reducers, status)) return false;
      w += (size_t)n * (1 + W);
    }
    if (!p.q.fixed) q_head++;
    if (nP != 0) return true;                  // 313-314: redraw while the pair set is empty
    if (p.q.fixed || p.q.no_redraw) return true;   // a fixed ideal with no pairs can never change; a listed one is its own (empty) run
  }
}

// lead-monomial observation, buchberger.cpp:354-370 + 391-394 / 403-406: row r = first k monomials of
// G[i] || first k of G[j] for the r-th pair, n exponents each, zero padded
template <int W, bool HASH = false>
__device__ uint64_t wave_obs(const Env<W>& e, const BbxParams& p, int env, int nP, bool write, bool want_hash) {
  const int lane = lane_id();
  const int n = p.nvars, k = p.k;
  const int cols = 2 * n * k;
  int32_t* out = (write && p.obs) ? p.obs + (size_t)env * p.obs_rows * cols : nullptr;
  const int rows = out ? (nP < p.obs_rows ? nP : p.obs_rows) : nP;
  const int items = rows * 2 * k;               // one item = one monomial slot of the matrix
  uint64_t h = 0;
  for (int it = lane; it < items; it += WAVE) {
    int r = it / (2 * k), rem = it - r * 2 * k;
    int half = rem / k, t = rem - half * k;
    uint32_t pr = e.pairs[r];
    int g = half ? (int)(pr >> 16) : (int)(pr & 0xffffu);
    bool have = t < (int)e.plen[g];
    Mono<W> mm = have ? e.am[e.poff[g] + t] : m_zero<W>();
    int base = it * n;
    if (out) obs_store<W>(out + base, mm, n);
    if (HASH && want_hash) for (int v = 0; v < n; v++) h += bbx_mix64((uint64_t)(base + v), m_exp(mm, v));
  }
  if (out && p.obs_fill) {
    for (int idx = rows * cols + lane; idx < p.obs_rows * cols; idx += WAVE) out[idx] = -1;
  }
  return (HASH && want_hash) ? wave_sum64(h) : 0;
}

template <int W, class EnvT>
__device__ uint64_t wave_pairs_hash(const EnvT& e, int nP) {
  uint64_t h = 0;
  for (int r = lane_id(); r < nP; r += WAVE) {
    uint32_t pr = e.pairs[r];
    h += bbx_mix64((uint64_t)(2 * r), pr & 0xffffu) + bbx_mix64((uint64_t)(2 * r + 1), pr >> 16);
  }
  return wave_sum64(h);
}
// words: [nterms, c0, e0[8], c1, e1[8], ...]  (oracle/trace.py poly_words)
template <int W>
__device__ uint64_t wave_poly_hash(const Env<W>& e, int g) {
  const int n = e.plen[g], off = e.poff[g];
  uint64_t h = lane_id() == 0 ? bbx_mix64(0, (uint32_t)n) : 0;
  for (int t = lane_id(); t < n; t += WAVE) {
    Mono<W> mm = e.am[off + t];
    uint64_t b = 1 + (uint64_t)t * 9;
    h += bbx_mix64(b, e.ac[off + t]);
    for (int v = 0; v < BBX_MAXVARS; v++) h += bbx_mix64(b + 1 + v, v < 2 * W - 1 ? m_exp(mm, v) : 0u);
  }
  return wave_sum64(h);
}

// ------------------------------------------------------------------ selection strategies (buchberger.cpp:165-240)
// Index of the pair with the minimal key (First/Degree/Normal/Sugar) or the maximal one (Last/Codegree/Strange/
// Spice).  P is always in ascending (j, i) order (new pairs carry the largest j and are appended sorted by i,
// removals keep the order), so the row index IS the reference's (j, i) tie-break in both directions.
// Keys: Degree = deg lcm; Normal = lcm in grevlex; Sugar = (sugar of the pair, lcm).
template <int W> struct SelKey { uint32_t s; Mono<W> m; uint32_t r; };
template <int W> __device__ __forceinline__ bool sel_less(const SelKey<W>& a, const SelKey<W>& b) {
  if (a.s != b.s) return a.s < b.s;
  if (m_gt(b.m, a.m)) return true;
  if (m_gt(a.m, b.m)) return false;
  return a.r < b.r;
}
template <int W, class EnvT, class SugarFn>
__device__ __forceinline__ int select_pair_inl(const EnvT& e, int nP, int agent, SugarFn sugar_of) {
  const bool rev = agent >= BBX_AGENT_LAST;          // pick the maximum
  const bool by_deg = agent == BBX_AGENT_DEGREE || agent == BBX_AGENT_CODEGREE;
  const bool by_sugar = agent == BBX_AGENT_SUGAR || agent == BBX_AGENT_SPICE;
  SelKey<W> best; best.m = m_zero<W>();
  best.s = rev ? 0u : 0xFFFFFFFFu; best.r = rev ? 0u : 0xFFFFFFFFu;   // the identity of min / max: never beats a real row
  for (int r = lane_id(); r < nP; r += WAVE) {
    const uint32_t pr = e.pairs[r];
    const int i = pr & 0xffffu, j = pr >> 16;
    const Mono<W> li = e.lm[i], lj = e.lm[j];
    const Mono<W> l = m_lcm(li, lj);
    SelKey<W> c; c.r = (uint32_t)r; c.m = m_zero<W>(); c.s = 0;
    if (by_deg) c.s = m_deg(l);
    else {
      c.m = l;
      if (by_sugar) {
        const uint32_t si = (uint32_t)sugar_of(i) + m_deg(m_div(l, li)), sj = (uint32_t)sugar_of(j) + m_deg(m_div(l, lj));
        c.s = si > sj ? si : sj;
      }
    }
    if (rev ? sel_less<W>(best, c) : sel_less<W>(c, best)) best = c;
  }
  for (int o = 32; o > 0; o >>= 1) {
    SelKey<W> t;
    t.s = (uint32_t)__shfl_xor((int)best.s, o, WAVE); t.r = (uint32_t)__shfl_xor((int)best.r, o, WAVE);
#pragma unroll
    for (int q = 0; q < W; q++) t.m.w[q] = (uint32_t)__shfl_xor((int)best.m.w[q], o, WAVE);
    if (rev ? sel_less<W>(best, t) : sel_less<W>(t, best)) best = t;
  }
  return (int)best.r;
}
// (out of line where the inliner puts it there; environments that carry state in their view — BEnvC — take the inlined form,
// or the view would have to live in memory for the call)
template <int W, class EnvT, class SugarFn>
__device__ int select_pair(const EnvT& e, int nP, int agent, SugarFn sugar_of) { return select_pair_inl<W>(e, nP, agent, sugar_of); }
// choice(P.begin(), P.end(), rng) with std::default_random_engine (= minstd_rand0, x <- 16807 x mod 2^31-1) and a
// fresh std::uniform_int_distribution<>(0, n-1) per call (ideals.h:68-73): libstdc++'s downscaling branch, since
// the engine's range 2^31-3 always exceeds n-1: draw until below n*floor(range/n), then divide.
__device__ __forceinline__ int std_choice(uint32_t& x, int n) {
  const uint32_t urngrange = 2147483645u, uerange = (uint32_t)n;
  const uint32_t scaling = urngrange / uerange, past = uerange * scaling;
  uint32_t ret;
  if (x == 0u || x >= 2147483647u) x = 1u;      // never a state of the engine; guards the loop against a corrupt header
  do {
    x = (uint32_t)(((uint64_t)x * 16807u) % 2147483647u);
    ret = x - 1u;
  } while (ret >= past);
  return (int)(ret / scaling);
}
// discounted-return bookkeeping of value(): stats.discounted_return += discount * reward; discount *= gamma,
// in double and without fusing the multiply into the add (the reference runs on x86-64 without FMA)
__device__ __forceinline__ void value_accumulate(double& vret, double& vdisc, double reward, double gamma) {
#pragma clang fp contract(off)
  const double term = vdisc * reward;
  vret = vret + term;
  vdisc = vdisc * gamma;
}
// copy the live prefix of every persistent array between the HBM record and the LDS working copy
template <int W>
__device__ void stage_copy(const Env<W>& dst, const Env<W>& src, int nG, int nP, int nT) {
  const int lane = lane_id();
  for (int i = lane; i < nG; i += WAVE) {
    dst.lm[i] = src.lm[i]; dst.slm[i] = src.slm[i]; dst.sidx[i] = src.sidx[i]; dst.poff[i] = src.poff[i];
    dst.plen[i] = src.plen[i]; dst.psug[i] = src.psug[i]; dst.pinv[i] = src.pinv[i];
  }
  for (int i = lane; i < nP; i += WAVE) dst.pairs[i] = src.pairs[i];
  for (int i = lane; i < nT; i += WAVE) { dst.am[i] = src.am[i]; dst.ac[i] = src.ac[i]; }
}
