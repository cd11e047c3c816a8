// Batched polynomial algebra on the device (bbx_alg_* of include/bbx.h): the reference's free functions
//   Polynomial +, -, *            deepgroebner/polynomials.cpp:148-210
//   spoly / reduce / update       deepgroebner/buchberger.cpp:18-99
//   minimalize / interreduce      deepgroebner/buchberger.cpp:102-122
// on lists of polynomials that live in HBM.  A list is one record of the general layout (bbx_common.h): its polynomials
// are the record's basis elements (lead monomial, arena offset, length, sugar, 1/LC), its pair set the record's pair set.
// One wavefront per list; a batch of lists per launch.  Every operation works in the record's scratch polynomials and
// appends its result as a new element at the very end (or builds a fresh list in a second record), so a list that runs
// out of room is left exactly as it was: the host enlarges the records and runs the operation again for those lists.
#include "bbx_device.h"

enum { ALG_ADD = 0, ALG_SUB = 1, ALG_MUL = 2, ALG_SPOLY = 3, ALG_REDUCE = 4, ALG_UPDATE = 5, ALG_MINIMALIZE = 6, ALG_INTERREDUCE = 7 };

struct AlgParams {
  char* recs; char* recs2;            // the lists; the lists minimalize / interreduce build
  BbxLayout L;
  int32_t n, op, elim;
  const int32_t* args;                // [n][4] operands: element indices i, j (binary operations), dividend index and number of divisors (reduce)
  int32_t* out;                       // [n][4]: {status (0 = done), reduction steps, -, -}
};

// G.push_back(polynomial of n terms at sm / sc): arena copy and metadata; false (status set) when the list has no room
template <int W>
__device__ bool alg_append(const Env<W>& e, const BbxLayout& L, int& nG, int& arena_used, const Mono<W>* sm, const uint16_t* sc, int n, int sugar, int* status) {
  const int lane = lane_id();
  if (nG >= (int)L.maxG) { *status = BBX_ST_G_FULL; return false; }
  if (n > 65535) { *status = BBX_ST_POLY_LIMIT; return false; }
  if (arena_used + n > (int)L.arena) { *status = BBX_ST_ARENA_FULL; return false; }
  const int g = nG, off = arena_used;
  for (int t = lane; t < n; t += WAVE) { e.am[off + t] = sm[t]; e.ac[off + t] = sc[t]; }
  if (lane == 0) {
    e.poff[g] = (uint32_t)off; e.plen[g] = (uint16_t)n; e.psug[g] = (uint16_t)sugar;
    if (n > 0) { e.lm[g] = sm[0]; e.pinv[g] = (uint16_t)invmod(sc[0]); }
    else { e.lm[g] = m_zero<W>(); e.pinv[g] = 0; }
  }
  wave_sync();
  nG = g + 1; arena_used = off + n;
  return true;
}

template <int W>
__device__ int alg_merge(const PView<W>& A, const PView<W>& B, char* mlds, Mono<W>* tm, uint16_t* tc, Mono<W>* om, uint16_t* oc, int maxT) {
  if (A.n + B.n > 2 * maxT) return -1;
  const bool big = A.n > 0 && B.n > 0 && A.n + B.n > 64;
  return big ? wave_merge_tiled<W>(A, B, mlds, om, oc, maxT) : wave_merge<W>(A, B, tm, tc, om, oc, maxT);
}

// reduce(h, F) (buchberger.cpp:24-49) with F = the list's elements [0, nF) in list order, `skip` excepted (-1: none); h sits
// in scratch buffer 0 with hn terms and sugar hsug.  The remainder ends up in (rm, rc); returns its length or -1 (status).
template <int W>
__device__ int alg_reduce(const Env<W>& e, int nF, int skip, int hn, int hsug, int maxT, char* mlds, int* steps_out, int* rsug_out, int* status) {
  const int lane = lane_id();
  Mono<W>* hm0 = e.hm;            uint16_t* hc0 = e.hc;
  Mono<W>* hm1 = e.hm + maxT;     uint16_t* hc1 = e.hc + maxT;
  Mono<W>* rm = e.hm + 2 * maxT;  uint16_t* rc = e.hc + 2 * maxT;
  Mono<W>* tm = e.hm + 3 * maxT;  uint16_t* tc = e.hc + 3 * maxT;
  Mono<W>* hm = hm0; uint16_t* hc = hc0;
  int hoff = 0, steps = 0, rn = 0, rsug = 0;
  while (hn - hoff > 0) {
    const Mono<W> lmh = hm[hoff];
    int found = -1;
    for (int base = 0; base < nF; base += WAVE) {             // first divisor in list order (29-33)
      const int k = base + lane;
      const bool d = k < nF && k != skip && (int)e.plen[k] > 0 && m_divides(e.lm[k], lmh);
      const uint64_t mask = ballot64(d);
      if (mask) { found = base + __builtin_ctzll(mask); break; }
    }
    if (found >= 0) {                                          // h <- h - (LT h / LT f) f   (34-36)
      const int g = found;
      const uint32_t c = mulmod((uint32_t)uni((int)hc[hoff]), (uint32_t)uni((int)e.pinv[g]));
      const int offg = uni((int)e.poff[g]);
      PView<W> A, Bv;
      A.m = hm + hoff + 1; A.c = hc + hoff + 1; A.n = hn - hoff - 1; A.shift = m_zero<W>(); A.scale = 1;
      Bv.m = e.am + offg + 1; Bv.c = e.ac + offg + 1; Bv.n = uni((int)e.plen[g]) - 1;
      Bv.shift = m_div(lmh, e.lm[g]); Bv.scale = negmod(c);
      const int fs = uni((int)e.psug[g]) + (int)m_deg(Bv.shift);
      hsug = uni(fs > hsug ? fs : hsug);
      if (hsug > 65535) { *status = BBX_ST_DEG_OVERFLOW; return -1; }
      Mono<W>* nm = (hm == hm0) ? hm1 : hm0; uint16_t* nc = (hc == hc0) ? hc1 : hc0;
      const int nn = alg_merge<W>(A, Bv, mlds, tm, tc, nm, nc, maxT);
      if (nn < 0) { *status = BBX_ST_POLY_TOO_LONG; return -1; }
      hm = nm; hc = nc; hn = nn; hoff = 0;
      if (++steps > (1 << 24)) { *status = BBX_ST_RUNAWAY; return -1; }
    } else {                                                   // r <- r + LT h ; h <- h - LT h   (41-44)
      if (rn >= maxT) { *status = BBX_ST_POLY_TOO_LONG; return -1; }
      if (lane == 0) { rm[rn] = lmh; rc[rn] = hc[hoff]; }
      const int d = uni((int)m_deg(lmh));
      rsug = d > rsug ? d : rsug;
      rn++; hoff++;
    }
  }
  wave_sync();
  *steps_out = steps; *rsug_out = rsug > hsug ? rsug : hsug;
  return rn;
}

template <int W>
__global__ __launch_bounds__(256) void bbx_alg_kernel(AlgParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = lane_id();
  const int wave_in_block = uni((int)(threadIdx.x / WAVE));
  const int k = (int)(blockIdx.x * (blockDim.x / WAVE)) + wave_in_block;
  if (k >= p.n) return;
  if (p.out[4 * (size_t)k] == 0) return;                        // done in an earlier attempt (the host re-runs an operation after enlarging the records)
  char* rec = p.recs + (size_t)k * p.L.rec_bytes;
  BbxHdr* hdr = (BbxHdr*)rec;
  const BbxLayout& L = p.L;
  Env<W> e = env_view<W>(rec, L);
  char* mlds = smem + (size_t)wave_in_block * merge_lds_bytes<W>();
  const int maxT = (int)L.maxT;
  int nG = uni(hdr->nG), nP = uni(hdr->nP), arena_used = uni(hdr->arena_used);
  int status = BBX_ST_OK, steps = 0;
  const int a0 = uni(p.args[4 * (size_t)k]), a1 = uni(p.args[4 * (size_t)k + 1]);
  Mono<W>* hm0 = e.hm;            uint16_t* hc0 = e.hc;
  Mono<W>* hm1 = e.hm + maxT;     uint16_t* hc1 = e.hc + maxT;
  Mono<W>* rm = e.hm + 2 * maxT;  uint16_t* rc = e.hc + 2 * maxT;
  Mono<W>* tm = e.hm + 3 * maxT;  uint16_t* tc = e.hc + 3 * maxT;
  auto view = [&](int g, int skip_lead) {
    PView<W> v;
    const int off = uni((int)e.poff[g]);
    v.m = e.am + off + skip_lead; v.c = e.ac + off + skip_lead; v.n = uni((int)e.plen[g]) - skip_lead;
    v.shift = m_zero<W>(); v.scale = 1;
    return v;
  };

  if (p.op == ALG_ADD || p.op == ALG_SUB) {                     // polynomials.cpp:148-185
    PView<W> A = view(a0, 0), B = view(a1, 0);
    if (p.op == ALG_SUB) B.scale = BBX_P - 1u;                  // (-1) * g, then operator+
    const int n = alg_merge<W>(A, B, mlds, tm, tc, hm0, hc0, maxT);
    const int sa = uni((int)e.psug[a0]), sb = uni((int)e.psug[a1]);
    if (n < 0) status = BBX_ST_POLY_TOO_LONG;
    else alg_append<W>(e, L, nG, arena_used, hm0, hc0, n, sa > sb ? sa : sb, &status);
  } else if (p.op == ALG_SPOLY) {                               // buchberger.cpp:18-21
    const Mono<W> lmi = e.lm[a0], lmj = e.lm[a1];
    const Mono<W> gamma = m_lcm(lmi, lmj);
    PView<W> A = view(a0, 1), B = view(a1, 1);
    A.shift = m_div(gamma, lmi); A.scale = (uint32_t)uni((int)e.pinv[a0]);
    B.shift = m_div(gamma, lmj); B.scale = negmod((uint32_t)uni((int)e.pinv[a1]));
    const int si = uni((int)e.psug[a0]) + (int)m_deg(A.shift), sj = uni((int)e.psug[a1]) + (int)m_deg(B.shift);
    const int n = alg_merge<W>(A, B, mlds, tm, tc, hm0, hc0, maxT);
    if ((si > sj ? si : sj) > 65535) status = BBX_ST_DEG_OVERFLOW;
    else if (n < 0) status = BBX_ST_POLY_TOO_LONG;
    else alg_append<W>(e, L, nG, arena_used, hm0, hc0, n, si > sj ? si : sj, &status);
  } else if (p.op == ALG_MUL) {                                 // polynomials.cpp:205-210: g = g + t * f2 for every term t of f1
    const int offi = uni((int)e.poff[a0]), ni = uni((int)e.plen[a0]);
    Mono<W>* cm = hm0; uint16_t* cc = hc0;
    int cn = 0, sug = 0;
    for (int t = 0; t < ni && status == BBX_ST_OK; t++) {
      PView<W> A, B = view(a1, 0);
      A.m = cm; A.c = cc; A.n = cn; A.shift = m_zero<W>(); A.scale = 1;
      B.shift = e.am[offi + t]; B.scale = (uint32_t)uni((int)e.ac[offi + t]);
      const int st = (int)m_deg(B.shift) + uni((int)e.psug[a1]);
      sug = st > sug ? st : sug;
      Mono<W>* nm = (cm == hm0) ? hm1 : hm0; uint16_t* nc = (cc == hc0) ? hc1 : hc0;
      const int nn = alg_merge<W>(A, B, mlds, tm, tc, nm, nc, maxT);
      if (nn < 0) { status = BBX_ST_POLY_TOO_LONG; break; }
      cm = nm; cc = nc; cn = nn;
    }
    if (sug > 65535) status = BBX_ST_DEG_OVERFLOW;
    if (status == BBX_ST_OK) alg_append<W>(e, L, nG, arena_used, cm, cc, cn, sug, &status);
  } else if (p.op == ALG_REDUCE) {                              // buchberger.cpp:24-49: dividend = element a0, divisors = elements [0, a1)
    const int off = uni((int)e.poff[a0]), hn = uni((int)e.plen[a0]);
    if (hn > maxT) status = BBX_ST_POLY_TOO_LONG;
    else {
      for (int t = lane; t < hn; t += WAVE) { hm0[t] = e.am[off + t]; hc0[t] = e.ac[off + t]; }
      wave_sync();
      int rsug = 0;
      const int rn = alg_reduce<W>(e, a1, -1, hn, uni((int)e.psug[a0]), maxT, mlds, &steps, &rsug, &status);
      if (rn >= 0) alg_append<W>(e, L, nG, arena_used, rm, rc, rn, rsug, &status);
    }
  } else if (p.op == ALG_UPDATE) {                              // buchberger.cpp:52-99: f = the last element joins G = the elements before it
    int m = nG - 1;
    if (m < 0) status = BBX_ST_BAD_ACTION;
    else if (nP + m > (int)L.maxP) status = BBX_ST_P_FULL;      // (at most m new pairs: checked before anything is modified)
    else if (!wave_update<W>(e, L, m, nP, e.lm[m], p.elim, &status)) { /* status set */ }
  } else if (p.op == ALG_MINIMALIZE || p.op == ALG_INTERREDUCE) {
    // the result is a NEW list, built in the second record
    char* rec2 = p.recs2 + (size_t)k * L.rec_bytes;
    Env<W> e2 = env_view<W>(rec2, L);
    int nG2 = 0, arena2 = 0;
    if (p.op == ALG_MINIMALIZE) {                               // buchberger.cpp:102-111
      uint16_t* ord = (uint16_t*)e.lcm;                         // (the update's scratch)
      uint16_t* kept = ord + L.maxG;
      if (lane == 0) {
        for (int i = 0; i < nG; i++) ord[i] = (uint16_t)i;
        SortCtx<W> c{e.lm, ord};
        ss_std_sort<W>(c, nG);                                  // std::sort by lead monomial, ascending
      }
      wave_sync();
      int nk = 0;
      for (int r = 0; r < nG && status == BBX_ST_OK; r++) {
        const int g = uni((int)ord[r]);
        const Mono<W> lmg = e.lm[g];
        bool div = false;
        for (int base = 0; base < nk && !div; base += WAVE) {   // none_of(Gmin, LM f | LM g)
          const int q = base + lane;
          div = ballot64(q < nk && m_divides(e.lm[kept[q]], lmg)) != 0;
        }
        if (!div) {
          if (lane == 0) kept[nk] = (uint16_t)g;
          nk++;
          wave_sync();
          const int off = uni((int)e.poff[g]);
          alg_append<W>(e2, L, nG2, arena2, e.am + off, e.ac + off, uni((int)e.plen[g]), uni((int)e.psug[g]), &status);
        }
      }
    } else {                                                    // buchberger.cpp:114-122: t * (reduce(g - LT g, G) + LT g), t = 1 / LC g
      for (int g = 0; g < nG && status == BBX_ST_OK; g++) {
        const int off = uni((int)e.poff[g]), n = uni((int)e.plen[g]);
        if (n - 1 > maxT) { status = BBX_ST_POLY_TOO_LONG; break; }
        for (int t = lane; t < n - 1; t += WAVE) { hm0[t] = e.am[off + 1 + t]; hc0[t] = e.ac[off + 1 + t]; }
        wave_sync();
        int st = 0, rsug = 0;
        const int rn = alg_reduce<W>(e, nG, -1, n - 1, uni((int)e.psug[g]), maxT, mlds, &st, &rsug, &status);
        if (rn < 0) break;
        if (rn + 1 > maxT) { status = BBX_ST_POLY_TOO_LONG; break; }
        // [LT g] ++ r, every coefficient times 1 / LC g: staged in buffer 0 (the reduction is through with it)
        const uint32_t inv = (uint32_t)uni((int)e.pinv[g]);
        const Mono<W> lmg = e.lm[g];
        wave_sync();
        for (int t = lane; t < rn; t += WAVE) { hm0[1 + t] = rm[t]; hc0[1 + t] = (uint16_t)mulmod(rc[t], inv); }
        if (lane == 0) { hm0[0] = lmg; hc0[0] = 1; }
        wave_sync();
        const int dl = (int)m_deg(lmg);
        alg_append<W>(e2, L, nG2, arena2, hm0, hc0, rn + 1, rsug > dl ? rsug : dl, &status);
        steps += st;
      }
    }
    if (status == BBX_ST_OK && lane == 0) {
      BbxHdr h2 = *hdr;
      h2.nG = nG2; h2.nP = 0; h2.arena_used = arena2; h2.status = BBX_ST_OK;
      *(BbxHdr*)rec2 = h2;
    }
  }
  if (lane == 0) {
    if (status == BBX_ST_OK && p.op < ALG_MINIMALIZE) { hdr->nG = nG; hdr->nP = nP; hdr->arena_used = arena_used; }
    p.out[4 * (size_t)k] = status == BBX_ST_OK ? 0 : status;
    p.out[4 * (size_t)k + 1] = steps;
  }
}

// bbx_alg_from_envs: the bases of environments as polynomial lists, copied on the device.  Records of the general layout
// (general / wide classes) are lists already: live prefixes move over (relayout).  Records of the binomial layout are
// unpacked: element g = {lc x^lm[g], tc x^tm[g]} (tc == 0: one term).
#include "bbx_pmlp.h"
#include "bbx_binom.h"
template <int W>
__global__ void bbx_alg_from_envs_kernel(const char* src_recs, BbxLayout Ls, const int32_t* idx, int n, char* dst_recs, BbxLayout Ld) {
  const int k = blockIdx.x * (blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
  if (k >= n) return;
  char* s = const_cast<char*>(src_recs) + (size_t)idx[k] * Ls.rec_bytes;
  char* d = dst_recs + (size_t)k * Ld.rec_bytes;
  BbxHdr h = *(const BbxHdr*)s;
  const int lane = lane_id();
  Env<W> e = env_view<W>(d, Ld);
  int arena = 0;
  if (Ls.kind == 1) {
    const BEnv<W> b = benv_view<W>(s, Ls);
    for (int base = 0; base < h.nG; base += WAVE) {
      const int g = base + lane;
      uint2 gi = make_uint2(0, 0);
      if (g < h.nG) gi = b.ginfo[g];
      const int nt = g < h.nG ? ((gi.x >> 16) ? 2 : 1) : 0;
      int off = nt;                                           // exclusive prefix sum of the lengths over the wave
      for (int o = 1; o < WAVE; o <<= 1) { const int t = __shfl_up(off, o, WAVE); if (lane >= o) off += t; }
      const int total = __shfl(off, WAVE - 1, WAVE);
      off = arena + off - nt;
      if (g < h.nG) {
        e.lm[g] = b.lm[g]; e.poff[g] = (uint32_t)off; e.plen[g] = (uint16_t)nt; e.psug[g] = (uint16_t)(gi.y >> 16); e.pinv[g] = (uint16_t)(gi.y & 0xffffu);
        e.am[off] = b.lm[g]; e.ac[off] = (uint16_t)(gi.x & 0xffffu);
        if (nt == 2) { e.am[off + 1] = b.tm[g]; e.ac[off + 1] = (uint16_t)(gi.x >> 16); }
      }
      arena += total;
    }
  } else {
    stage_copy<W>(e, env_view<W>(s, Ls), h.nG, 0, h.arena_used);
    arena = h.arena_used;
  }
  if (lane == 0) {
    BbxHdr o = {};
    o.nG = h.nG; o.arena_used = arena;
    *(BbxHdr*)d = o;
  }
}
extern "C" int bbx_launch_alg_from_envs(const char* src_recs, const BbxLayout* Ls, const int32_t* idx, int n, char* dst_recs, const BbxLayout* Ld, hipStream_t stream) {
  const int blocks = (n + 3) / 4;
  if (Ls->W == 2) hipLaunchKernelGGL((bbx_alg_from_envs_kernel<2>), dim3(blocks), dim3(256), 0, stream, src_recs, *Ls, idx, n, dst_recs, *Ld);
  else if (Ls->W == 4) hipLaunchKernelGGL((bbx_alg_from_envs_kernel<4>), dim3(blocks), dim3(256), 0, stream, src_recs, *Ls, idx, n, dst_recs, *Ld);
  else hipLaunchKernelGGL((bbx_alg_from_envs_kernel<8>), dim3(blocks), dim3(256), 0, stream, src_recs, *Ls, idx, n, dst_recs, *Ld);
  return (int)hipGetLastError();
}

extern "C" int bbx_launch_alg(const AlgParams* p, hipStream_t stream) {
  const int waves = 4, blocks = (p->n + waves - 1) / waves;
#define BBX_ALG_LAUNCH(WW) do { \
    const size_t lds = (size_t)waves * merge_lds_bytes<WW>(); \
    hipError_t err_ = hipFuncSetAttribute((const void*)bbx_alg_kernel<WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (err_ != hipSuccess) return (int)err_; \
    hipLaunchKernelGGL((bbx_alg_kernel<WW>), dim3(blocks), dim3(waves * WAVE), lds, stream, *p); } while (0)
  if (p->L.W == 2) BBX_ALG_LAUNCH(2); else if (p->L.W == 4) BBX_ALG_LAUNCH(4); else BBX_ALG_LAUNCH(8);
#undef BBX_ALG_LAUNCH
  return (int)hipGetLastError();
}
