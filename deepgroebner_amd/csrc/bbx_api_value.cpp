// libbbx.so — LeadMonomialsEnv::value (buchberger.cpp:332-351): discounted returns of full Buchberger rollouts from clones of
// the current states (bbx_value, bbx_values, bbx_values_seeded of include/bbx.h).
#include <algorithm>
#include <cstring>
#include <vector>

#include "bbx_batch.h"

using namespace bbx_host;

// ---- value(): discounted return of full Buchberger rollouts from clones of the current states ---------------------
namespace {

int agent_of_strategy(const char* s) {   // unknown keys select First: std::map::operator[] default (buchberger.cpp:342-349)
  if (!strcmp(s, "degree")) return BBX_AGENT_DEGREE;
  if (!strcmp(s, "normal")) return BBX_AGENT_NORMAL;
  if (!strcmp(s, "sugar")) return BBX_AGENT_SUGAR;
  if (!strcmp(s, "random")) return BBX_AGENT_STDRANDOM;   // choice(P, rng) of a seeded std::default_random_engine (buchberger.cpp:200-203, 244)
  return BBX_AGENT_FIRST;
}

// std::default_random_engine::seed(s) (linear_congruential_engine<uint_fast32_t, 16807, 0, 2^31-1>, libstdc++ bits/random.tcc):
// the int seed converts to the unsigned result type first; x = s mod m, and 0 becomes 1
uint32_t minstd_state_of_seed(long long seed) {
  const uint32_t x = (uint32_t)((uint64_t)seed % 2147483647ull);
  return x ? x : 1u;
}

// One rollout to completion per entry of src (indices into b), from clones of the current states; seeds != null: the
// engine states of the clones' seeded Random selection.  Everything runs on the default stream without a copy in between
// and with one wait: values + completion marks come back in one transfer at the end (clones whose generator lead monomials
// tie get the reducer order buchberger()'s std::sort would give them from a kernel: bbx_value_resort_kernel).  3-variable binomial batches run on the register/LDS-resident class (bbx_fast_value_kernel),
// environments that outgrow it and every other batch on the HBM-resident class of the batch, long-polynomial
// environments one workgroup per clone.  A clone that runs out of room enlarges the records of the whole batch
// (grow_records) and the rollouts start again.
int value_rollouts(bbx_batch* b, const std::vector<int32_t>& src, int agent, const std::vector<uint32_t>* seeds, double gamma, double* out) {
  const int n = (int)src.size();
  if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
  for (int attempt = 0; attempt < 40; attempt++) {
    if (n > b->vcap) {
      void* old[] = {b->d_vrecs, b->d_vhdr, b->d_vsrc, b->d_vseeds, b->d_vvals};
      for (void* q : old) (void)hipFree(q);
      b->d_vrecs = nullptr; b->d_vhdr = nullptr; b->d_vsrc = nullptr; b->d_vseeds = nullptr; b->d_vvals = nullptr; b->vcap = 0;
      HIPCHK(hipMalloc((void**)&b->d_vrecs, (size_t)n * b->L.rec_bytes));
      HIPCHK(hipMalloc((void**)&b->d_vhdr, (size_t)n));                        // clone flags (u8)
      HIPCHK(hipMalloc((void**)&b->d_vsrc, (size_t)n * sizeof(int32_t)));
      HIPCHK(hipMalloc((void**)&b->d_vseeds, (size_t)n * sizeof(uint32_t)));
      HIPCHK(hipMalloc((void**)&b->d_vvals, (size_t)n * 2 * sizeof(double)));  // {value, completion mark} per clone
      b->vcap = n;
    }
    HIPCHK(hipMemcpyAsync(b->d_vsrc, src.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, 0));
    if (seeds) HIPCHK(hipMemcpyAsync(b->d_vseeds, seeds->data(), (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, 0));
    const int ngen = b->sort_reducers ? b->gens[0]->npolys() : 0;
    uint8_t* d_flags = (uint8_t*)b->d_vhdr;
    int lrc = bbx_launch_clone(b->d_recs, b->d_vrecs, &b->L, b->d_vsrc, nullptr, n, seeds ? b->d_vseeds : nullptr, 0, 1, ngen, d_flags, 0);
    if (lrc) return fail(BBX_E_DEVICE, "clone launch failed: %s", hipGetErrorString((hipError_t)lrc));
    lrc = bbx_launch_value_resort(b->d_vrecs, &b->L, n, d_flags, 0);   // (clones whose generators tie: std::sort's order)
    if (lrc) return fail(BBX_E_DEVICE, "resort launch failed: %s", hipGetErrorString((hipError_t)lrc));
    BbxParams p; fill_params(b, &p);
    p.recs = b->d_vrecs; p.B = n; p.nsteps = 1 << 30; p.set_budget = 1; p.agent = agent; p.auto_reset = 0;
    p.value_mode = 1; p.gamma = gamma; p.values = nullptr; p.trace = nullptr; p.accounting = 0;
    p.lite = nullptr;                                           // the clones are not the batch's environments
    if (b->wide) lrc = bbx_launch_step(&p, 4, b->wide, 0);
    else {
      const bool vfast = b->fast && b->staged;                  // (every selection strategy: bbx_fast.h, f_select_ordered)
      lrc = 0;
      if (vfast) lrc = bbx_launch_step(&p, 3, b->envs_per_block, 0);
      if (b->gen_to_wide) p.spill_terms = 384;
      if (!lrc) { if (vfast) { p.set_budget = 0; p.pass = 1; } lrc = bbx_launch_step(&p, 0, b->envs_per_block, 0); }
      if (!lrc && b->gen_to_wide) { p.set_budget = 0; p.pass = 1; p.spill_terms = 0; lrc = bbx_launch_step(&p, 4, 8, 0); }
    }
    if (lrc) return fail(BBX_E_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    lrc = bbx_launch_value_collect(b->d_vrecs, b->L.rec_bytes, n, b->d_vvals, 0);
    if (lrc) return fail(BBX_E_DEVICE, "collect launch failed: %s", hipGetErrorString((hipError_t)lrc));
    std::vector<double> v2((size_t)n * 2);
    HIPCHK(hipMemcpy(v2.data(), b->d_vvals, v2.size() * sizeof(double), hipMemcpyDeviceToHost));   // the only wait of the call
    unsigned grow = 0; int grow_k = -1;
    for (int k = 0; k < n; k++) {
      const int st = (int)v2[2 * (size_t)k + 1];
      if (st == 0) continue;
      if (st > 0 && bbx_st_capacity(st) && !b->no_growth) { grow |= 1u << st; if (grow_k < 0) grow_k = k; continue; }
      return fail(BBX_E_CAPACITY, "value rollout of environment %d did not finish: %s", src[k], st > 0 ? status_name(st) : "pairs left");
    }
    if (!grow) {
      for (int k = 0; k < n; k++) out[k] = v2[2 * (size_t)k];
      return BBX_OK;
    }
    int rc = grow_records(b, grow, src[grow_k], 0);             // (frees the clones: sized by the old layout)
    if (rc) return rc;
  }
  return fail(BBX_E_CAPACITY, "value rollouts kept outgrowing the records");
}

// `seeds`: explicit seeds of the Random rollouts — [n] for "random", [n][100] for "sample" — or null: drawn from the
// handle's own stream (the reference seeds from std::random_device: ours starts from the handle's seed base, so a run is
// reproducible under BBX_DEFAULT_SEED)
int values_for(bbx_batch* b, const std::vector<int32_t>& envs, const char* strategy, double gamma, const int64_t* seeds, double* out) {
  const int n = (int)envs.size();
  auto draw = [b]() { return (long long)(b->value_rng() & 0x7fffffffull); };
  if (!strcmp(strategy, "sample")) {          // best of one Degree and 100 Random rollouts (buchberger.cpp:333-341)
    int rc = value_rollouts(b, envs, BBX_AGENT_DEGREE, nullptr, gamma, out);
    if (rc) return rc;
    std::vector<int32_t> src; std::vector<uint32_t> st;
    src.reserve((size_t)n * 100); st.reserve((size_t)n * 100);
    for (int k = 0; k < n; k++) for (int i = 0; i < 100; i++) { src.push_back(envs[k]); st.push_back(minstd_state_of_seed(seeds ? seeds[(size_t)k * 100 + i] : draw())); }
    std::vector<double> r(src.size());
    rc = value_rollouts(b, src, BBX_AGENT_STDRANDOM, &st, gamma, r.data());
    if (rc) return rc;
    for (int k = 0; k < n; k++) for (int i = 0; i < 100; i++) out[k] = std::max(out[k], r[(size_t)k * 100 + i]);
    return BBX_OK;
  }
  const int agent = agent_of_strategy(strategy);
  if (agent == BBX_AGENT_STDRANDOM) {
    std::vector<uint32_t> st(n);
    for (int k = 0; k < n; k++) st[k] = minstd_state_of_seed(seeds ? seeds[k] : draw());
    return value_rollouts(b, envs, agent, &st, gamma, out);
  }
  return value_rollouts(b, envs, agent, nullptr, gamma, out);
}

}  // namespace

extern "C" int bbx_value(bbx_batch* b, int idx, const char* strategy, double gamma, double* out) {
  if (!b || !strategy || !out || idx < 0 || idx >= b->B) return fail(BBX_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  return values_for(b, std::vector<int32_t>{idx}, strategy, gamma, nullptr, out);
}

extern "C" int bbx_values_seeded(bbx_batch* b, const char* strategy, double gamma, const int64_t* seeds, double* out) {
  if (!b || !strategy || !out) return fail(BBX_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  std::vector<int32_t> envs(b->B);
  for (int e = 0; e < b->B; e++) envs[e] = e;
  return values_for(b, envs, strategy, gamma, seeds, out);
}

extern "C" int bbx_values(bbx_batch* b, const char* strategy, double gamma, double* out) {
  if (!b || !strategy || !out) return fail(BBX_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  std::vector<int32_t> envs(b->B);
  for (int e = 0; e < b->B; e++) envs[e] = e;
  return values_for(b, envs, strategy, gamma, nullptr, out);
}

