// libbbx.so — introspection (statistics, states, reduced bases, traces), the ideal generators on their own and the text format
// of ideals (include/bbx.h).
#include <algorithm>
#include <cstring>
#include <vector>

#include "bbx_batch.h"

using namespace bbx_host;

extern "C" {

int bbx_stats(bbx_batch* b, int64_t* out8) {
  int64_t* out6 = out8;
  if (!b || !out6) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->ps_active) { int rc_ = session_close(b, false, nullptr, false); if (rc_) return rc_; }
  HIPCHK(hipDeviceSynchronize());
  int rc = read_headers(b);
  if (rc) return rc;
  for (int e = 0; e < b->B; e++) {
    const BbxHdr& h = b->h_hdr[e];
    int64_t* o = out6 + (size_t)e * 8;
    o[0] = h.total_steps; o[1] = h.total_additions; o[2] = h.episodes; o[3] = h.zero_reductions; o[4] = h.status; o[5] = h.q_head;
    o[6] = h.alg_bytes; o[7] = h.nG;
  }
  return BBX_OK;
}

// (bbx_alg.cpp) where the records of a quiet batch live
int bbx_internal_records(bbx_batch* b, const char** recs, BbxLayout* L, int* device, int* W, int* batch) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
  HIPCHK(hipDeviceSynchronize());
  *recs = b->d_recs; *L = b->L; *device = b->device; *W = b->W; *batch = b->B;
  return BBX_OK;
}

int bbx_capacities(bbx_batch* b, int32_t* out5) {
  if (!b || !out5) return fail(BBX_E_ARG, "null argument");
  out5[0] = (int32_t)b->L.maxG; out5[1] = (int32_t)b->L.maxP; out5[2] = (int32_t)b->L.arena; out5[3] = (int32_t)b->L.maxT; out5[4] = b->grow_events;
  return BBX_OK;
}

int bbx_env_status(bbx_batch* b, int32_t* status) {
  if (!b || !status) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->ps_active) { int rc_ = session_close(b, false, nullptr, false); if (rc_) return rc_; }
  HIPCHK(hipDeviceSynchronize());
  int rc = read_headers(b);
  if (rc) return rc;
  for (int e = 0; e < b->B; e++) status[e] = b->h_hdr[e].status;
  return BBX_OK;
}

int bbx_state_sizes(bbx_batch* b, int idx, int32_t* basis_size, int32_t* npairs, int32_t* nterms_total) {
  if (!b || idx < 0 || idx >= b->B) return fail(BBX_E_ARG, "bad environment index");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->ps_active) { int rc_ = session_close(b, false, nullptr, false); if (rc_) return rc_; }
  HIPCHK(hipDeviceSynchronize());
  BbxHdr h;
  HIPCHK(hipMemcpy(&h, b->d_recs + (size_t)idx * b->L.rec_bytes, sizeof h, hipMemcpyDeviceToHost));
  if (basis_size) *basis_size = h.nG;
  if (npairs) *npairs = h.nP;
  if (nterms_total) *nterms_total = b->binom ? 2 * h.nG : h.arena_used;   // binomial class: upper bound
  return BBX_OK;
}

int bbx_state_get(bbx_batch* b, int idx, int32_t* nterms, int32_t* coefs, int32_t* exps, int32_t* pairs, int32_t* order) {
  if (!b || idx < 0 || idx >= b->B) return fail(BBX_E_ARG, "bad environment index");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->ps_active) { int rc_ = session_close(b, false, nullptr, false); if (rc_) return rc_; }
  HIPCHK(hipDeviceSynchronize());
  const char* rec = b->d_recs + (size_t)idx * b->L.rec_bytes;
  BbxHdr h;
  HIPCHK(hipMemcpy(&h, rec, sizeof h, hipMemcpyDeviceToHost));
  if (b->binom) {
    const int W = b->W, nG = h.nG, nP = h.nP;
    std::vector<uint32_t> lm((size_t)std::max(nG, 1) * W), tm((size_t)std::max(nG, 1) * W), gi((size_t)std::max(nG, 1) * 2),
        si((size_t)std::max(nG, 1) * 2), pr(std::max(nP, 1));
    if (nG) {
      HIPCHK(hipMemcpy(lm.data(), rec + b->L.off_lm, (size_t)nG * W * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(tm.data(), rec + b->L.off_tm, (size_t)nG * W * 4, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(gi.data(), rec + b->L.off_ginfo, (size_t)nG * 8, hipMemcpyDeviceToHost));
      HIPCHK(hipMemcpy(si.data(), rec + b->L.off_sinfo, (size_t)nG * 8, hipMemcpyDeviceToHost));
    }
    if (nP) HIPCHK(hipMemcpy(pr.data(), rec + b->L.off_pairs, (size_t)nP * 4, hipMemcpyDeviceToHost));
    auto unpack = [W](const uint32_t* w, int32_t* e8) {
      for (int v = 0; v < bbx::kN; v++)
        e8[v] = v < 2 * W - 1 ? ((v & 1) ? (int)(w[v >> 1] >> 16) : (int)(w[v >> 1] & 0xffffu)) : 0;
    };
    size_t at = 0;
    for (int g = 0; g < nG; g++) {
      const uint32_t c0 = gi[2 * g] & 0xffffu, c1 = gi[2 * g] >> 16;
      if (nterms) nterms[g] = c1 ? 2 : 1;
      if (coefs) coefs[at] = (int)c0;
      if (exps) unpack(lm.data() + (size_t)g * W, exps + at * bbx::kN);
      at++;
      if (c1) {
        if (coefs) coefs[at] = (int)c1;
        if (exps) unpack(tm.data() + (size_t)g * W, exps + at * bbx::kN);
        at++;
      }
      if (order) order[g] = (int)(si[2 * g + 1] >> 16);
    }
    if (pairs) for (int r = 0; r < nP; r++) { pairs[2 * r] = (int)(pr[r] & 0xffffu); pairs[2 * r + 1] = (int)(pr[r] >> 16); }
    return BBX_OK;
  }
  const int W = b->W, nG = h.nG, nP = h.nP, nt = h.arena_used;
  std::vector<uint32_t> am((size_t)std::max(nt, 1) * W), poff(std::max(nG, 1)), pr(std::max(nP, 1));
  std::vector<uint16_t> ac(std::max(nt, 1)), plen(std::max(nG, 1)), sidx(std::max(nG, 1));
  if (nt) {
    HIPCHK(hipMemcpy(am.data(), rec + b->L.off_am, (size_t)nt * W * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ac.data(), rec + b->L.off_ac, (size_t)nt * 2, hipMemcpyDeviceToHost));
  }
  if (nG) {
    HIPCHK(hipMemcpy(poff.data(), rec + b->L.off_poff, (size_t)nG * 4, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(plen.data(), rec + b->L.off_plen, (size_t)nG * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(sidx.data(), rec + b->L.off_sidx, (size_t)nG * 2, hipMemcpyDeviceToHost));
  }
  if (nP) HIPCHK(hipMemcpy(pr.data(), rec + b->L.off_pairs, (size_t)nP * 4, hipMemcpyDeviceToHost));
  size_t at = 0;
  for (int g = 0; g < nG; g++) {
    if (nterms) nterms[g] = plen[g];
    for (int t = 0; t < plen[g]; t++, at++) {
      const uint32_t* w = am.data() + ((size_t)poff[g] + t) * W;
      if (coefs) coefs[at] = ac[poff[g] + t];
      if (exps) {
        for (int v = 0; v < bbx::kN; v++) {
          int x = 0;
          if (v < 2 * W - 1) x = (v & 1) ? (int)(w[v >> 1] >> 16) : (int)(w[v >> 1] & 0xffffu);
          exps[at * bbx::kN + v] = x;
        }
      }
    }
    if (order) order[g] = sidx[g];
  }
  if (pairs) for (int r = 0; r < nP; r++) { pairs[2 * r] = (int)(pr[r] & 0xffffu); pairs[2 * r + 1] = (int)(pr[r] >> 16); }
  return BBX_OK;
}

// interreduce(minimalize(G)) of environment idx's current basis (what buchberger() returns, buchberger.cpp:265), computed on
// the device (bbx_alg_from_envs + bbx_alg_minimalize + bbx_alg_interreduce).  Two-call protocol like bbx_state_get: sizes
// first (nterms == NULL), then the data.
int bbx_reduced_basis(bbx_batch* b, int idx, int32_t* basis_size, int32_t* nterms_total, int32_t* nterms, int32_t* coefs, int32_t* exps) {
  if (!b || idx < 0 || idx >= b->B) return fail(BBX_E_ARG, "bad environment index");
  bbx_alg* a = nullptr;
  const int32_t e = idx;
  int rc = bbx_alg_from_envs(b, 1, &e, &a);                 // the basis as a device-resident list, then the two kernels
  if (!rc) rc = bbx_alg_minimalize(a);
  if (!rc) rc = bbx_alg_interreduce(a);
  int32_t n = 0, tot = 0;
  if (!rc) rc = bbx_alg_sizes(a, &n, &tot);
  if (!rc) {
    if (basis_size) *basis_size = n;
    if (nterms_total) *nterms_total = tot;
    if (nterms) rc = bbx_alg_get(a, 0, nterms, coefs, exps, nullptr);
  }
  bbx_alg_destroy(a);
  return rc;
}

int bbx_trace_enable(bbx_batch* b, int capacity_steps) {
  if (!b || capacity_steps < 0) return fail(BBX_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  HIPCHK(hipDeviceSynchronize());
  if (b->d_trace) { HIPCHK(hipFree(b->d_trace)); b->d_trace = nullptr; }
  b->trace_cap = capacity_steps;
  if (capacity_steps) {
    HIPCHK(hipMalloc((void**)&b->d_trace, (size_t)b->B * capacity_steps * sizeof(BbxTraceRec)));
    HIPCHK(hipMemset(b->d_trace, 0, (size_t)b->B * capacity_steps * sizeof(BbxTraceRec)));
  }
  return BBX_OK;
}

int bbx_trace_read(bbx_batch* b, int env, int first, int count, bbx_trace_rec* out) {
  if (!b || !out || !b->d_trace || env < 0 || env >= b->B || first < 0 || count < 0 || first + count > b->trace_cap)
    return fail(BBX_E_ARG, "bad trace range");
  static_assert(sizeof(bbx_trace_rec) == sizeof(BbxTraceRec), "trace record layouts must match");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  HIPCHK(hipMemcpy(out, b->d_trace + (size_t)env * b->trace_cap + first, (size_t)count * sizeof(BbxTraceRec), hipMemcpyDeviceToHost));
  return BBX_OK;
}

// ---- generators on their own -------------------------------------------------------------------
int bbx_gen_create(const char* ideal_dist, bbx_gen** out) {
  if (!ideal_dist || !out) return fail(BBX_E_ARG, "null argument");
  std::string err;
  auto g = bbx::parse_ideal_dist(ideal_dist, &err);
  if (!g) return fail(BBX_E_ARG, "%s", err.c_str());
  *out = new bbx_gen{std::move(g), {}};
  return BBX_OK;
}
void bbx_gen_destroy(bbx_gen* g) { delete g; }
int bbx_gen_seed(bbx_gen* g, int64_t seed) { if (!g) return fail(BBX_E_ARG, "null"); g->g->seed(seed); return BBX_OK; }
int bbx_gen_nvars(const bbx_gen* g) { return g ? g->g->nvars() : 0; }
int bbx_gen_next(bbx_gen* g, int32_t* npolys, int32_t* nterms_total) {
  if (!g) return fail(BBX_E_ARG, "null");
  std::string err;
  if (!g->g->next(g->last, &err)) return fail(BBX_E_GENERATOR, "%s", err.c_str());
  int tot = 0;
  for (auto& f : g->last) tot += (int)f.t.size();
  if (npolys) *npolys = (int)g->last.size();
  if (nterms_total) *nterms_total = tot;
  return BBX_OK;
}
int bbx_gen_get(const bbx_gen* g, int32_t* nterms, int32_t* coefs, int32_t* exps, int32_t* sugars) {
  if (!g) return fail(BBX_E_ARG, "null");
  size_t at = 0;
  for (size_t p = 0; p < g->last.size(); p++) {
    const auto& f = g->last[p];
    if (nterms) nterms[p] = (int)f.t.size();
    if (sugars) sugars[p] = f.sugar;
    for (auto& t : f.t) {
      if (coefs) coefs[at] = t.c;
      if (exps) for (int v = 0; v < bbx::kN; v++) exps[at * bbx::kN + v] = t.e[v];
      at++;
    }
  }
  return BBX_OK;
}

int bbx_parse_ideal(const char* text, int32_t cap_polys, int32_t cap_terms, int32_t* npolys, int32_t* nterms_total,
                    int32_t* nterms, int32_t* coefs, int32_t* exps) {
  if (!text || !npolys || !nterms_total) return fail(BBX_E_ARG, "null argument");
  bbx::HIdeal F; std::string err;
  if (!bbx::parse_ideal_string(text, F, &err)) return fail(BBX_E_ARG, "%s", err.c_str());
  size_t total = 0;
  for (auto& f : F) total += f.t.size();
  *npolys = (int32_t)F.size(); *nterms_total = (int32_t)total;
  if (!nterms && !coefs && !exps) return BBX_OK;             // size query
  if ((int64_t)F.size() > cap_polys || (int64_t)total > cap_terms) return fail(BBX_E_CAPACITY, "output buffers too small: %zu polynomials, %zu terms", F.size(), total);
  size_t at = 0;
  for (size_t p = 0; p < F.size(); p++) {
    if (nterms) nterms[p] = (int32_t)F[p].t.size();
    for (auto& t : F[p].t) {
      if (coefs) coefs[at] = t.c;
      if (exps) for (int v = 0; v < bbx::kN; v++) exps[at * bbx::kN + v] = t.e[v];
      at++;
    }
  }
  return BBX_OK;
}

int bbx_format_ideal(int npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps, char* out, int cap) {
  if (npolys < 0 || (npolys && (!nterms || !coefs || !exps)) || cap < 0 || (cap && !out)) return fail(BBX_E_ARG, "bad argument");
  std::string s;
  size_t at = 0;
  for (int p = 0; p < npolys; p++) {
    std::vector<bbx::HTerm> ts;
    for (int k = 0; k < nterms[p]; k++, at++) {
      bbx::HTerm t; t.c = bbx::coef_norm(coefs[at]); t.deg = 0;
      for (int v = 0; v < bbx::kN; v++) { t.e[v] = exps[at * bbx::kN + v]; if (t.e[v] < 0) return fail(BBX_E_ARG, "negative exponent"); t.deg += t.e[v]; }
      if (t.c) ts.push_back(t);
    }
    if (p) s += '|';
    s += bbx::format_polynomial(bbx::poly_from_terms(ts));
  }
  if ((int)s.size() + 1 <= cap) memcpy(out, s.c_str(), s.size() + 1);
  return (int)s.size();                                          // length needed, excluding the terminator
}

}  // extern "C"
