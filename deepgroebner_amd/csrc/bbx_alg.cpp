// libbbx.so — batched polynomial algebra on the device: the host side of bbx_alg_* (include/bbx.h).  Lists of polynomials
// are marshalled into records of the general layout (bbx_common.h) and handed to the kernels of bbx_algebra.hip.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "bbx_host.h"
#include "bbx_ideals.h"

struct AlgParams {                    // (bbx_algebra.hip)
  char* recs; char* recs2;
  BbxLayout L;
  int32_t n, op, elim;
  const int32_t* args;
  int32_t* out;
};
extern "C" int bbx_launch_alg(const AlgParams* p, hipStream_t stream);
extern "C" int bbx_launch_relayout(const char* src_recs, char* dst_recs, const BbxLayout* Ls, const BbxLayout* Ld, int B, hipStream_t stream);
extern "C" int bbx_launch_alg_from_envs(const char* src_recs, const BbxLayout* Ls, const int32_t* idx, int n, char* dst_recs, const BbxLayout* Ld, hipStream_t stream);
// (bbx_api.cpp) quiesces the batch and reports where its records live
extern "C" int bbx_internal_records(bbx_batch* b, const char** recs, BbxLayout* L, int* device, int* W, int* batch);
extern "C" int bbx_launch_gather_hdr(const char* recs, uint32_t rec_bytes, int B, BbxHdr* out, hipStream_t stream);

using bbx_host::fail;

struct bbx_alg {
  int n = 0, device = 0, W = 2;
  BbxLayout L{};
  char* d_recs = nullptr; char* d_recs2 = nullptr;
  int32_t* d_args = nullptr; int32_t* d_out = nullptr; BbxHdr* d_hdr = nullptr;
  std::vector<BbxHdr> hdr;            // host mirror of the lists' headers (sizes)
  ~bbx_alg() {
    (void)hipSetDevice(device);
    void* dev[] = {d_recs, d_recs2, d_args, d_out, d_hdr};
    for (void* q : dev) if (q) (void)hipFree(q);
  }
};

namespace {

int pack_mono(int W, const bbx::HTerm& t, uint32_t* w) {
  const int slots = 2 * W;
  uint32_t s[16] = {0};
  for (int v = 0; v < bbx::kN; v++) {
    if (t.e[v] == 0) continue;
    if (t.e[v] < 0 || t.e[v] > 65535) return fail(BBX_E_UNSUPPORTED, "exponent %d out of range", t.e[v]);
    s[v] = (uint32_t)t.e[v];
  }
  if (t.deg > 65535) return fail(BBX_E_UNSUPPORTED, "degree %d out of range", t.deg);
  s[slots - 1] = (uint32_t)t.deg;
  for (int i = 0; i < W; i++) w[i] = s[2 * i] | (s[2 * i + 1] << 16);
  return BBX_OK;
}

int refresh_headers(bbx_alg* a) {
  a->hdr.resize(a->n);
  int lrc = bbx_launch_gather_hdr(a->d_recs, a->L.rec_bytes, a->n, a->d_hdr, 0);
  if (lrc) return fail(BBX_E_DEVICE, "gather launch failed: %s", hipGetErrorString((hipError_t)lrc));
  HIPCHK(hipMemcpy(a->hdr.data(), a->d_hdr, (size_t)a->n * sizeof(BbxHdr), hipMemcpyDeviceToHost));
  return BBX_OK;
}

// enlarge the records (what was full doubles) and move the lists over
int grow(bbx_alg* a, unsigned need) {
  uint64_t maxG = a->L.maxG, maxP = a->L.maxP, arena = a->L.arena, maxT = a->L.maxT;
  if (need & (1u << BBX_ST_G_FULL)) maxG *= 2;
  if (need & (1u << BBX_ST_P_FULL)) maxP *= 2;
  if (need & (1u << BBX_ST_ARENA_FULL)) arena *= 2;
  if (need & (1u << BBX_ST_POLY_TOO_LONG)) maxT *= 2;
  while (arena < 2 * maxT) arena *= 2;
  while (maxP < 2 * maxG) maxP *= 2;
  const uint64_t MW = 4ull * a->W;
  if (maxG > 65534 || maxT > (1u << 22) || 128ull + (3 * MW + 13) * maxG + 4 * maxP + (MW + 2) * (arena + 5 * maxT) > 0xE0000000ull)
    return fail(BBX_E_CAPACITY, "a polynomial list outgrew what a record can hold");
  const BbxLayout NL = bbx_host::make_layout(a->W, (int)maxG, (int)maxP, (int)arena, (int)maxT);
  char* nrecs = nullptr; char* nrecs2 = nullptr;
  HIPCHK(hipMalloc((void**)&nrecs, (size_t)a->n * NL.rec_bytes));
  HIPCHK(hipMalloc((void**)&nrecs2, (size_t)a->n * NL.rec_bytes));
  int lrc = bbx_launch_relayout(a->d_recs, nrecs, &a->L, &NL, a->n, 0);
  if (lrc) return fail(BBX_E_DEVICE, "relayout launch failed: %s", hipGetErrorString((hipError_t)lrc));
  HIPCHK(hipDeviceSynchronize());
  (void)hipFree(a->d_recs); (void)hipFree(a->d_recs2);
  a->d_recs = nrecs; a->d_recs2 = nrecs2; a->L = NL;
  return BBX_OK;
}

// one operation over all lists; lists that run out of room get larger records and the operation again
int run(bbx_alg* a, int op, int elim, const int32_t* args4, int32_t* steps) {
  HIPCHK(hipSetDevice(a->device));
  std::vector<int32_t> out((size_t)a->n * 4, -1);
  if (args4) HIPCHK(hipMemcpy(a->d_args, args4, (size_t)a->n * 4 * sizeof(int32_t), hipMemcpyHostToDevice));
  else HIPCHK(hipMemset(a->d_args, 0, (size_t)a->n * 4 * sizeof(int32_t)));
  HIPCHK(hipMemcpy(a->d_out, out.data(), out.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  const bool rebuilds = op >= 6;                         // minimalize / interreduce build new lists in the second record array
  for (int attempt = 0; attempt < 64; attempt++) {
    AlgParams p{};
    p.recs = a->d_recs; p.recs2 = a->d_recs2; p.L = a->L; p.n = a->n; p.op = op; p.elim = elim; p.args = a->d_args; p.out = a->d_out;
    int lrc = bbx_launch_alg(&p, 0);
    if (lrc) return fail(BBX_E_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    HIPCHK(hipMemcpy(out.data(), a->d_out, out.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    unsigned need = 0;
    for (int k = 0; k < a->n; k++) {
      const int st = out[(size_t)k * 4];
      if (st == 0) continue;
      if (bbx_st_capacity(st)) { need |= 1u << st; continue; }
      return fail(st == BBX_ST_BAD_ACTION ? BBX_E_ARG : BBX_E_CAPACITY, "list %d: the operation failed (status %d)", k, st);
    }
    if (!need) {
      if (rebuilds) std::swap(a->d_recs, a->d_recs2);
      if (steps) for (int k = 0; k < a->n; k++) steps[k] = out[(size_t)k * 4 + 1];
      return refresh_headers(a);
    }
    int rc = grow(a, need);
    if (rc) return rc;
    if (rebuilds) {                                      // (the new lists of the others went with the old second array: all again)
      std::fill(out.begin(), out.end(), -1);
      HIPCHK(hipMemcpy(a->d_out, out.data(), out.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
  }
  return fail(BBX_E_CAPACITY, "the operation kept outgrowing the records");
}

int check_index(const bbx_alg* a, int k, int i) {
  if (i < 0 || i >= a->hdr[k].nG) return fail(BBX_E_ARG, "list %d has no element %d", k, i);
  return BBX_OK;
}

}  // namespace

extern "C" {

int bbx_alg_create(int device, int nlists, const int32_t* npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps, bbx_alg** out) {
  if (!out) return fail(BBX_E_ARG, "out is null");
  *out = nullptr;
  if (nlists < 1 || !npolys) return fail(BBX_E_ARG, "bad polynomial lists");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BBX_E_DEVICE, "no HIP device available (libbbx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(BBX_E_DEVICE, "device %d out of range (have %d)", device, ndev);
  HIPCHK(hipSetDevice(device));
  // the lists as polynomials (Polynomial's constructor: terms sorted descending, sugar = degree of the lead term)
  std::vector<std::vector<bbx::HPoly>> lists(nlists);
  size_t pi = 0, at = 0;
  int maxvar = 0, max_polys = 0; size_t max_terms_total = 0, max_terms = 0;
  for (int k = 0; k < nlists; k++) {
    size_t total = 0;
    for (int q = 0; q < npolys[k]; q++, pi++) {
      std::vector<bbx::HTerm> ts;
      for (int t = 0; t < nterms[pi]; t++, at++) {
        bbx::HTerm h; h.c = bbx::coef_norm(coefs[at]); h.deg = 0;
        for (int v = 0; v < bbx::kN; v++) { h.e[v] = exps[at * bbx::kN + v]; h.deg += h.e[v]; if (h.e[v]) maxvar = std::max(maxvar, v + 1); }
        if (h.c == 0) return fail(BBX_E_ARG, "list %d: a term with coefficient 0 (the reference's polynomials never hold one)", k);
        ts.push_back(h);
      }
      lists[k].push_back(ts.empty() ? bbx::HPoly() : bbx::poly_from_terms(ts));
      total += ts.size(); max_terms = std::max(max_terms, ts.size());
    }
    max_polys = std::max(max_polys, npolys[k]); max_terms_total = std::max(max_terms_total, total);
  }
  auto a = std::make_unique<bbx_alg>();
  a->n = nlists; a->device = device;
  a->W = maxvar <= 3 ? 2 : (maxvar <= 7 ? 4 : 8);
  const int maxG = std::max(8, (max_polys + 4 + 1) & ~1);
  const int maxT = (int)std::max<size_t>(64, 2 * max_terms);
  const int arena = (int)std::max<size_t>(2 * (size_t)maxT, 2 * max_terms_total + 64);
  a->L = bbx_host::make_layout(a->W, maxG, std::max(64, 2 * maxG), arena, maxT);
  const BbxLayout& L = a->L;
  std::vector<char> img((size_t)nlists * L.rec_bytes, 0);
  for (int k = 0; k < nlists; k++) {
    char* rec = img.data() + (size_t)k * L.rec_bytes;
    BbxHdr* h = (BbxHdr*)rec;
    uint32_t off = 0;
    for (size_t g = 0; g < lists[k].size(); g++) {
      const bbx::HPoly& f = lists[k][g];
      ((uint32_t*)(rec + L.off_poff))[g] = off;
      ((uint16_t*)(rec + L.off_plen))[g] = (uint16_t)f.t.size();
      ((uint16_t*)(rec + L.off_psug))[g] = (uint16_t)f.sugar;
      ((uint16_t*)(rec + L.off_pinv))[g] = f.t.empty() ? 0 : (uint16_t)bbx::coef_inv(f.t[0].c);
      if (f.t.size() > 65535) return fail(BBX_E_UNSUPPORTED, "a polynomial with more than 65535 terms");
      for (size_t t = 0; t < f.t.size(); t++) {
        int rc = pack_mono(a->W, f.t[t], (uint32_t*)(rec + L.off_am) + (size_t)(off + t) * a->W);
        if (rc) return rc;
        ((uint16_t*)(rec + L.off_ac))[off + t] = (uint16_t)f.t[t].c;
      }
      if (!f.t.empty()) memcpy(rec + L.off_lm + g * 4 * a->W, rec + L.off_am + (size_t)off * 4 * a->W, 4 * a->W);
      off += (uint32_t)f.t.size();
    }
    h->nG = (int32_t)lists[k].size(); h->arena_used = (int32_t)off;
  }
  HIPCHK(hipMalloc((void**)&a->d_recs, img.size()));
  HIPCHK(hipMalloc((void**)&a->d_recs2, img.size()));
  HIPCHK(hipMalloc((void**)&a->d_args, (size_t)nlists * 4 * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&a->d_out, (size_t)nlists * 4 * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&a->d_hdr, (size_t)nlists * sizeof(BbxHdr)));
  HIPCHK(hipMemcpy(a->d_recs, img.data(), img.size(), hipMemcpyHostToDevice));
  int rc = refresh_headers(a.get());
  if (rc) return rc;
  *out = a.release();
  return BBX_OK;
}

void bbx_alg_destroy(bbx_alg* a) { delete a; }

int bbx_alg_from_envs(bbx_batch* b, int n, const int32_t* envs, bbx_alg** out) {
  if (!b || !out || n < 1) return fail(BBX_E_ARG, "bad arguments");
  *out = nullptr;
  const char* recs = nullptr; BbxLayout Ls{}; int device = 0, W = 2, batch = 0;
  int rc = bbx_internal_records(b, &recs, &Ls, &device, &W, &batch);
  if (rc) return rc;
  std::vector<int32_t> idx(n);
  for (int k = 0; k < n; k++) {
    idx[k] = envs ? envs[k] : k;
    if (idx[k] < 0 || idx[k] >= batch) return fail(BBX_E_ARG, "environment index out of range");
  }
  HIPCHK(hipSetDevice(device));
  auto a = std::make_unique<bbx_alg>();
  a->n = n; a->device = device; a->W = W;
  // room for every basis as it is (binomial layout: two terms per element), scratch for interreduce's reductions
  const int maxG = (int)Ls.maxG, arena = Ls.kind == 1 ? 2 * (int)Ls.maxG + 64 : (int)Ls.arena;
  const int maxT = Ls.kind == 1 ? 64 : (int)std::max<uint32_t>(64u, std::min<uint32_t>(Ls.maxT, 4096u));
  a->L = bbx_host::make_layout(W, maxG, std::max(64, 2 * maxG), std::max(arena, 2 * maxT), maxT);
  HIPCHK(hipMalloc((void**)&a->d_recs, (size_t)n * a->L.rec_bytes));
  HIPCHK(hipMalloc((void**)&a->d_recs2, (size_t)n * a->L.rec_bytes));
  HIPCHK(hipMalloc((void**)&a->d_args, (size_t)n * 4 * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&a->d_out, (size_t)n * 4 * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&a->d_hdr, (size_t)n * sizeof(BbxHdr)));
  HIPCHK(hipMemcpy(a->d_args, idx.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  int lrc = bbx_launch_alg_from_envs(recs, &Ls, a->d_args, n, a->d_recs, &a->L, 0);
  if (lrc) return fail(BBX_E_DEVICE, "copy launch failed: %s", hipGetErrorString((hipError_t)lrc));
  HIPCHK(hipDeviceSynchronize());
  rc = refresh_headers(a.get());
  if (rc) return rc;
  *out = a.release();
  return BBX_OK;
}

int bbx_alg_binop(bbx_alg* a, int op, const int32_t* ij) {
  if (!a || !ij || op < 0 || op > 3) return fail(BBX_E_ARG, "bad arguments");
  std::vector<int32_t> args((size_t)a->n * 4, 0);
  for (int k = 0; k < a->n; k++) {
    for (int s = 0; s < 2; s++) { int rc = check_index(a, k, ij[2 * k + s]); if (rc) return rc; args[(size_t)k * 4 + s] = ij[2 * k + s]; }
    if (op == 3 && (a->hdr[k].nG < 1)) return fail(BBX_E_ARG, "list %d is empty", k);
  }
  return run(a, op, 0, args.data(), nullptr);
}

int bbx_alg_reduce(bbx_alg* a, const int32_t* dividend_and_ndivisors, int32_t* steps) {
  if (!a || !dividend_and_ndivisors) return fail(BBX_E_ARG, "bad arguments");
  std::vector<int32_t> args((size_t)a->n * 4, 0);
  for (int k = 0; k < a->n; k++) {
    const int g = dividend_and_ndivisors[2 * k], nF = dividend_and_ndivisors[2 * k + 1];
    int rc = check_index(a, k, g);
    if (rc) return rc;
    if (nF < 0 || nF > a->hdr[k].nG) return fail(BBX_E_ARG, "list %d: %d divisors of %d elements", k, nF, a->hdr[k].nG);
    args[(size_t)k * 4] = g; args[(size_t)k * 4 + 1] = nF;
  }
  return run(a, 4, 0, args.data(), steps);
}

int bbx_alg_update(bbx_alg* a, int elimination, const int32_t* npairs, const int32_t* pairs, int32_t* npairs_out, int32_t* pairs_out, int pairs_cap) {
  if (!a || !npairs || !npairs_out || elimination < 0 || elimination > 2) return fail(BBX_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(a->device));
  size_t at = 0, total = 0;
  for (int k = 0; k < a->n; k++) total += (size_t)std::max(0, npairs[k]);
  if (total && !pairs) return fail(BBX_E_ARG, "pairs is null");
  for (int k = 0; k < a->n; k++) {                          // the pair sets go into the records (they may have to grow first)
    if (a->hdr[k].nG < 1) return fail(BBX_E_ARG, "list %d is empty: no polynomial to add", k);
    while ((uint32_t)(npairs[k] + a->hdr[k].nG) > a->L.maxP) { int rc = grow(a, 1u << BBX_ST_P_FULL); if (rc) return rc; }
  }
  for (int k = 0; k < a->n; k++) {
    std::vector<uint32_t> pr(std::max(1, npairs[k]));
    for (int r = 0; r < npairs[k]; r++, at++) {
      const int i = pairs[2 * at], j = pairs[2 * at + 1];
      if (i < 0 || j < 0 || i >= a->hdr[k].nG - 1 || j >= a->hdr[k].nG - 1) return fail(BBX_E_ARG, "list %d: pair (%d, %d) outside the basis", k, i, j);
      pr[r] = (uint32_t)i | ((uint32_t)j << 16);
    }
    char* rec = a->d_recs + (size_t)k * a->L.rec_bytes;
    if (npairs[k]) HIPCHK(hipMemcpy(rec + a->L.off_pairs, pr.data(), (size_t)npairs[k] * 4, hipMemcpyHostToDevice));
    const int32_t np = npairs[k];
    HIPCHK(hipMemcpy(rec + offsetof(BbxHdr, nP), &np, sizeof np, hipMemcpyHostToDevice));
  }
  int rc = run(a, 5, elimination, nullptr, nullptr);
  if (rc) return rc;
  at = 0;
  for (int k = 0; k < a->n; k++) {
    const int np = a->hdr[k].nP;
    npairs_out[k] = np;
    if (pairs_out) {
      if ((long long)at + np > pairs_cap) return fail(BBX_E_CAPACITY, "pairs_out holds %d pairs, more are needed", pairs_cap);
      std::vector<uint32_t> pr(std::max(1, np));
      if (np) HIPCHK(hipMemcpy(pr.data(), a->d_recs + (size_t)k * a->L.rec_bytes + a->L.off_pairs, (size_t)np * 4, hipMemcpyDeviceToHost));
      for (int r = 0; r < np; r++, at++) { pairs_out[2 * at] = (int32_t)(pr[r] & 0xffffu); pairs_out[2 * at + 1] = (int32_t)(pr[r] >> 16); }
    }
  }
  return BBX_OK;
}

int bbx_alg_minimalize(bbx_alg* a) { return a ? run(a, 6, 0, nullptr, nullptr) : fail(BBX_E_ARG, "null argument"); }
int bbx_alg_interreduce(bbx_alg* a) { return a ? run(a, 7, 0, nullptr, nullptr) : fail(BBX_E_ARG, "null argument"); }

int bbx_alg_sizes(bbx_alg* a, int32_t* npolys, int32_t* nterms_total) {
  if (!a) return fail(BBX_E_ARG, "null argument");
  for (int k = 0; k < a->n; k++) { if (npolys) npolys[k] = a->hdr[k].nG; if (nterms_total) nterms_total[k] = a->hdr[k].arena_used; }
  return BBX_OK;
}

int bbx_alg_get(bbx_alg* a, int list, int32_t* nterms, int32_t* coefs, int32_t* exps, int32_t* sugars) {
  if (!a || list < 0 || list >= a->n) return fail(BBX_E_ARG, "bad list index");
  HIPCHK(hipSetDevice(a->device));
  const BbxLayout& L = a->L;
  const char* rec = a->d_recs + (size_t)list * L.rec_bytes;
  const int nG = a->hdr[list].nG, nt = a->hdr[list].arena_used, W = a->W;
  std::vector<uint32_t> am((size_t)std::max(nt, 1) * W), poff(std::max(nG, 1));
  std::vector<uint16_t> ac(std::max(nt, 1)), plen(std::max(nG, 1)), psug(std::max(nG, 1));
  if (nt) {
    HIPCHK(hipMemcpy(am.data(), rec + L.off_am, (size_t)nt * W * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ac.data(), rec + L.off_ac, (size_t)nt * 2, hipMemcpyDeviceToHost));
  }
  if (nG) {
    HIPCHK(hipMemcpy(poff.data(), rec + L.off_poff, (size_t)nG * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(plen.data(), rec + L.off_plen, (size_t)nG * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(psug.data(), rec + L.off_psug, (size_t)nG * 2, hipMemcpyDeviceToHost));
  }
  size_t at = 0;
  for (int g = 0; g < nG; g++) {
    if (nterms) nterms[g] = plen[g];
    if (sugars) sugars[g] = psug[g];
    for (int t = 0; t < plen[g]; t++, at++) {
      const uint32_t* w = am.data() + ((size_t)poff[g] + t) * W;
      if (coefs) coefs[at] = ac[poff[g] + t];
      if (exps) for (int v = 0; v < bbx::kN; v++) exps[at * bbx::kN + v] = v < 2 * W - 1 ? ((v & 1) ? (int)(w[v >> 1] >> 16) : (int)(w[v >> 1] & 0xffffu)) : 0;
    }
  }
  return BBX_OK;
}

}  // extern "C"
