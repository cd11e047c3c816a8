// TEST INFRASTRUCTURE — not part of the product.
//
// C-ABI shim around the *unmodified* reference C++ sources, which are compiled
// where they lie under /root/reference (see oracle/Makefile; nothing from the
// reference is copied into this repository).  It exists so that
//   * oracle/bbx_oracle.c (our CPU restatement) can be pinned against the real
//     reference on identical inputs,
//   * tests/golden/ vectors can be generated (oracle/make_golden.py),
//   * bench.py can time the real reference (cpu_baseline.kind == "reference").
// The built library lands in oracle/_ref/ (git-ignored, travels with gpurun).
//
// The function set mirrors oracle/bbx_oracle.h one-to-one with a ref_ prefix,
// so the same Python harness (oracle/ffi.py) drives both.

#include <algorithm>
#include <array>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <optional>
#include <random>
#include <sstream>
#include <string>
#include <vector>

#include "buchberger.h"
#include "ideals.h"
#include "polynomials.h"

// BuchbergerEnv keeps its reducer list G_ and its ideal generator private.  The driver must read the former (to dump the
// reducer order into the golden traces) and set the latter (environments over a fixed ideal).  It reaches them through
// pointers to members obtained in explicit template instantiations — the one place where the language does not apply
// access checks to names ([temp.spec]) — so the reference headers are compiled exactly as they are: no macro games, no
// second definition of its classes.
namespace access {
template <typename Tag> struct Member { static typename Tag::type ptr; };
template <typename Tag> typename Tag::type Member<Tag>::ptr;
template <typename Tag, typename Tag::type P> struct Bind {
  struct Init { Init() { Member<Tag>::ptr = P; } };
  static Init init;
};
template <typename Tag, typename Tag::type P> typename Bind<Tag, P>::Init Bind<Tag, P>::init;
struct Reducers { typedef std::vector<Polynomial> BuchbergerEnv::*type; };
struct Generator { typedef std::unique_ptr<IdealGenerator> BuchbergerEnv::*type; };
template struct Bind<Reducers, &BuchbergerEnv::G_>;
template struct Bind<Generator, &BuchbergerEnv::ideal_gen>;
}  // namespace access
static std::vector<Polynomial>& reducers_of(BuchbergerEnv& env) { return env.*access::Member<access::Reducers>::ptr; }
static std::unique_ptr<IdealGenerator>& generator_of(BuchbergerEnv& env) { return env.*access::Member<access::Generator>::ptr; }

namespace {

struct PolyList {
  std::vector<Polynomial> v;
};

Polynomial make_poly(int nterms, const int* coef, const int* exps) {
  if (nterms == 0) return Polynomial{};
  std::vector<Term> t;
  for (int k = 0; k < nterms; k++) {
    std::array<int, N> e{};
    for (int x = 0; x < N; x++) e[x] = exps[k * N + x];
    t.push_back(Term{Coefficient{coef[k]}, Monomial{e}});
  }
  return Polynomial{t};
}

int coef_value(Coefficient c) {
  std::ostringstream os;
  os << c;
  return std::stoi(os.str());
}

struct Env {
  BuchbergerEnv env;
  Env(const std::string& d, EliminationType e, RewardType r, bool si, bool sr) : env{d, e, r, si, sr} {}
};

EliminationType elim_of(int e) {
  return e == 0 ? EliminationType::GebauerMoeller : (e == 1 ? EliminationType::LCM : EliminationType::None);
}

}  // namespace

extern "C" {

// ---------------------------------------------------------------- poly lists
void* ref_pl_new() { return new PolyList; }
void ref_pl_free(void* p) { delete static_cast<PolyList*>(p); }
void ref_pl_clear(void* p) { static_cast<PolyList*>(p)->v.clear(); }
int ref_pl_len(void* p) { return (int)static_cast<PolyList*>(p)->v.size(); }
void ref_pl_add(void* p, int nterms, const int* coef, const int* exps) {
  static_cast<PolyList*>(p)->v.push_back(make_poly(nterms, coef, exps));
}
int ref_pl_nterms(void* p, int i) { return static_cast<PolyList*>(p)->v[i].size(); }
int ref_pl_sugar(void* p, int i) { return static_cast<PolyList*>(p)->v[i].sugar(); }
void ref_pl_get(void* p, int i, int* coef, int* exps) {
  const Polynomial& f = static_cast<PolyList*>(p)->v[i];
  for (int k = 0; k < f.size(); k++) {
    coef[k] = coef_value(f.terms[k].coeff);
    for (int x = 0; x < N; x++) exps[k * N + x] = f.terms[k].monom[x];
  }
}

// ------------------------------------------------------------ field / monomial
int ref_coef_norm(int a) { return coef_value(Coefficient{a}); }
int ref_coef_add(int a, int b) { return coef_value(Coefficient{a} + Coefficient{b}); }
int ref_coef_sub(int a, int b) { return coef_value(Coefficient{a} - Coefficient{b}); }
int ref_coef_mul(int a, int b) { return coef_value(Coefficient{a} * Coefficient{b}); }
int ref_coef_div(int a, int b) { return coef_value(Coefficient{a} / Coefficient{b}); }
int ref_mono_gt(const int* a, const int* b) {
  std::array<int, N> x{}, y{};
  for (int i = 0; i < N; i++) { x[i] = a[i]; y[i] = b[i]; }
  return Monomial{x} > Monomial{y};
}

// ------------------------------------------------------------ core functions
void ref_poly_add(void* pl, int i, int j, void* out) {
  auto& v = static_cast<PolyList*>(pl)->v;
  static_cast<PolyList*>(out)->v.push_back(v[i] + v[j]);
}
void ref_poly_sub(void* pl, int i, int j, void* out) {
  auto& v = static_cast<PolyList*>(pl)->v;
  static_cast<PolyList*>(out)->v.push_back(v[i] - v[j]);
}
void ref_poly_mul(void* pl, int i, int j, void* out) {
  auto& v = static_cast<PolyList*>(pl)->v;
  static_cast<PolyList*>(out)->v.push_back(v[i] * v[j]);
}
void ref_parse_polynomial(const char* s, void* out) {
  static_cast<PolyList*>(out)->v.push_back(parse_polynomial(std::string(s)));
}
void ref_spoly(void* pl, int i, int j, void* out) {
  auto& v = static_cast<PolyList*>(pl)->v;
  static_cast<PolyList*>(out)->v.push_back(spoly(v[i], v[j]));
}
int ref_reduce(void* plg, int gi, void* plF, void* out) {
  auto [r, st] = reduce(static_cast<PolyList*>(plg)->v[gi], static_cast<PolyList*>(plF)->v);
  static_cast<PolyList*>(out)->v.push_back(r);
  return st.steps;
}
// pairs: int[2*cap] (i,j interleaved); returns new pair count
int ref_update(void* plG, int* pairs, int npairs, void* plf, int fi, int elim) {
  std::vector<SPair> P;
  for (int k = 0; k < npairs; k++) P.push_back(SPair{pairs[2 * k], pairs[2 * k + 1]});
  Polynomial f = static_cast<PolyList*>(plf)->v[fi];
  update(static_cast<PolyList*>(plG)->v, P, f, elim_of(elim));
  for (size_t k = 0; k < P.size(); k++) { pairs[2 * k] = P[k].i; pairs[2 * k + 1] = P[k].j; }
  return (int)P.size();
}
void ref_minimalize(void* pl, void* out) {
  static_cast<PolyList*>(out)->v = minimalize(static_cast<PolyList*>(pl)->v);
}
void ref_interreduce(void* pl, void* out) {
  static_cast<PolyList*>(out)->v = interreduce(static_cast<PolyList*>(pl)->v);
}
// stats: double[5] = zero_reductions, nonzero_reductions, polynomial_additions, total_reward, discounted_return
// npairs < 0  => start from the generators (the F-only overload)
void ref_buchberger(void* plF, const int* pairs, int npairs, int selection, int elim, int rewards,
                    int sort_input, int sort_reducers, double gamma, int has_seed, int seed,
                    void* out, double* stats) {
  std::optional<int> sd = has_seed ? std::optional<int>(seed) : std::nullopt;
  auto sel = static_cast<SelectionType>(selection);
  auto rw = rewards == 0 ? RewardType::Additions : RewardType::Reductions;
  std::pair<std::vector<Polynomial>, BuchbergerStats> res;
  if (npairs < 0) {
    res = buchberger(static_cast<PolyList*>(plF)->v, sel, elim_of(elim), rw, sort_input, sort_reducers, gamma, sd);
  } else {
    std::vector<SPair> S;
    for (int k = 0; k < npairs; k++) S.push_back(SPair{pairs[2 * k], pairs[2 * k + 1]});
    res = buchberger(static_cast<PolyList*>(plF)->v, S, sel, elim_of(elim), rw, sort_reducers, gamma, sd);
  }
  if (out) static_cast<PolyList*>(out)->v = res.first;
  stats[0] = res.second.zero_reductions;
  stats[1] = res.second.nonzero_reductions;
  stats[2] = res.second.polynomial_additions;
  stats[3] = res.second.total_reward;
  stats[4] = res.second.discounted_return;
}

// ------------------------------------------------------------ ideal generators
void ref_cyclic(int n, void* out) { static_cast<PolyList*>(out)->v = cyclic(n); }
int ref_basis(int n, int d, int* exps, int cap) {
  auto B = basis(n, d);
  for (size_t k = 0; k < B.size() && (int)k < cap; k++)
    for (int x = 0; x < N; x++) exps[k * N + x] = B[k][x];
  return (int)B.size();
}
int ref_degree_distribution(int n, int d, int dist, int constants, double* probs) {
  auto dd = degree_distribution(n, d, static_cast<DistributionType>(dist), constants);
  auto p = dd.probabilities();
  for (size_t k = 0; k < p.size(); k++) probs[k] = p[k];
  return (int)p.size();
}
void* ref_gen_new(const char* dist) { return parse_ideal_dist(dist).release(); }
void ref_gen_free(void* g) { delete static_cast<IdealGenerator*>(g); }
void ref_gen_seed(void* g, int seed) { static_cast<IdealGenerator*>(g)->seed(seed); }
int ref_gen_nvars(void* g) { return static_cast<IdealGenerator*>(g)->nvars(); }
void ref_gen_next(void* g, void* out) { static_cast<PolyList*>(out)->v = static_cast<IdealGenerator*>(g)->next(); }
void* ref_gen_copy(void* g) { return static_cast<IdealGenerator*>(g)->copy().release(); }

// ------------------------------------------------------------ BuchbergerEnv
void* ref_env_new(const char* dist, int elim, int rewards, int sort_input, int sort_reducers) {
  return new Env(dist, elim_of(elim), rewards == 0 ? RewardType::Additions : RewardType::Reductions,
                 sort_input, sort_reducers);
}
// env over a fixed ideal (FixedIdealGenerator), as the reference Python tests use
void* ref_env_new_fixed(void* pl, int elim, int rewards, int sort_input, int sort_reducers) {
  Env* e = new Env("cyclic-3", elim_of(elim), rewards == 0 ? RewardType::Additions : RewardType::Reductions,
                   sort_input, sort_reducers);
  generator_of(e->env) = std::make_unique<FixedIdealGenerator>(static_cast<PolyList*>(pl)->v);
  return e;
}
void ref_env_free(void* e) { delete static_cast<Env*>(e); }
void* ref_env_copy(void* e) { return new Env(*static_cast<Env*>(e)); }
void ref_env_seed(void* e, int seed) { static_cast<Env*>(e)->env.seed(seed); }
int ref_env_nvars(void* e) { return static_cast<Env*>(e)->env.nvars(); }
void ref_env_reset(void* e) { static_cast<Env*>(e)->env.reset(); }
double ref_env_step_pair(void* e, int i, int j) { return static_cast<Env*>(e)->env.step(SPair{i, j}); }
double ref_env_step(void* e, int action) {
  BuchbergerEnv& env = static_cast<Env*>(e)->env;
  return env.step(env.P[action]);
}
double ref_env_value(void* e, const char* strategy, double gamma) {
  return static_cast<Env*>(e)->env.value(strategy, gamma);
}
int ref_env_nG(void* e) { return (int)static_cast<Env*>(e)->env.G.size(); }
int ref_env_nP(void* e) { return (int)static_cast<Env*>(e)->env.P.size(); }
void ref_env_pairs(void* e, int* out) {
  auto& P = static_cast<Env*>(e)->env.P;
  for (size_t k = 0; k < P.size(); k++) { out[2 * k] = P[k].i; out[2 * k + 1] = P[k].j; }
}
int ref_env_poly_nterms(void* e, int i) { return static_cast<Env*>(e)->env.G[i].size(); }
int ref_env_poly_sugar(void* e, int i) { return static_cast<Env*>(e)->env.G[i].sugar(); }
void ref_env_poly_get(void* e, int i, int* coef, int* exps) {
  const Polynomial& f = static_cast<Env*>(e)->env.G[i];
  for (int k = 0; k < f.size(); k++) {
    coef[k] = coef_value(f.terms[k].coeff);
    for (int x = 0; x < N; x++) exps[k * N + x] = f.terms[k].monom[x];
  }
}
// reducer order: out[r] = index into G of the r-th reducer.  The reference
// keeps copies in G_; identical copies are matched to G in insertion order
// (equal lead monomials keep insertion order under upper_bound).
void ref_env_reducer_order(void* e, int* out) {
  BuchbergerEnv& env = static_cast<Env*>(e)->env;
  std::vector<char> used(env.G.size(), 0);
  const std::vector<Polynomial>& G_ = reducers_of(env);
  for (size_t r = 0; r < G_.size(); r++) {
    out[r] = -1;
    for (size_t i = 0; i < env.G.size(); i++) {
      if (!used[i] && env.G[i] == G_[r]) { used[i] = 1; out[r] = (int)i; break; }
    }
  }
}
// lead-monomial observation, exactly as LeadMonomialsEnv builds it
// (buchberger.cpp:354-408): rows = |P|, cols = 2*n*k
void ref_env_obs(void* e, int k, int n, int* out) {
  BuchbergerEnv& env = static_cast<Env*>(e)->env;
  int w = n * k;
  int row = 0;
  for (const auto& p : env.P) {
    std::vector<int> a = lead_monomials_vector(env.G[p.i], k, n);
    std::vector<int> b = lead_monomials_vector(env.G[p.j], k, n);
    std::copy(a.begin(), a.end(), out + row * 2 * w);
    std::copy(b.begin(), b.end(), out + row * 2 * w + w);
    row++;
  }
}

// ------------------------------------------------------------ LeadMonomialsEnv (the real wrapped class)
void* ref_lme_new(const char* dist, int sort_input, int sort_reducers, int k) {
  return new LeadMonomialsEnv(dist, sort_input, sort_reducers, k);
}
void ref_lme_free(void* e) { delete static_cast<LeadMonomialsEnv*>(e); }
void* ref_lme_copy(void* e) { return new LeadMonomialsEnv(*static_cast<LeadMonomialsEnv*>(e)); }
void ref_lme_seed(void* e, int seed) { static_cast<LeadMonomialsEnv*>(e)->seed(seed); }
void ref_lme_reset(void* e) { static_cast<LeadMonomialsEnv*>(e)->reset(); }
double ref_lme_step(void* e, int action) { return static_cast<LeadMonomialsEnv*>(e)->step(action); }
double ref_lme_value(void* e, const char* strategy, double gamma) {
  return static_cast<LeadMonomialsEnv*>(e)->value(strategy, gamma);
}
int ref_lme_cols(void* e) { return static_cast<LeadMonomialsEnv*>(e)->cols; }
int ref_lme_state_size(void* e) { return (int)static_cast<LeadMonomialsEnv*>(e)->state.size(); }
void ref_lme_state(void* e, int* out) {
  auto& s = static_cast<LeadMonomialsEnv*>(e)->state;
  std::copy(s.begin(), s.end(), out);
}

// ------------------------------------------------------------ CPU baseline timing
// Runs `nenvs` independent BuchbergerEnv instances (env e seeded seed0+e), each
// for `nsteps` steps with auto-reset, choosing the action with the same
// counter-based hash the device agent uses (bbx_agent_hash in include/bbx.h):
// action = (hash(agent_seed0+e, t) * |P|) >> 32.  The lead-monomial observation is
// rebuilt every step like LeadMonomialsEnv::step does.  Returns seconds; writes
// total steps and total additions (= -sum reward).
static inline uint32_t agent_hash(uint32_t seed, uint32_t t) {
  uint64_t z = ((uint64_t)seed << 32 | t) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
double ref_bench_random(const char* dist, int k, int nenvs, int nsteps, int seed0, int agent_seed0,
                        long long* total_steps, long long* total_additions, unsigned long long* checksum) {
  long long steps = 0, adds = 0;
  unsigned long long cs = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (int e = 0; e < nenvs; e++) {
    LeadMonomialsEnv env{dist, false, true, k};
    env.seed(seed0 + e);
    env.reset();
    for (int t = 0; t < nsteps; t++) {
      int rows = (int)env.state.size() / env.cols;
      int action = (int)(((uint64_t)agent_hash((uint32_t)(agent_seed0 + e), (uint32_t)t) * (uint32_t)rows) >> 32);
      double r = env.step(action);
      steps++;
      adds += (long long)(-r);
      cs = cs * 1000003ull + (unsigned long long)(env.state.size() * 31 + (long long)(-r));
      if (env.state.empty()) env.reset();
    }
  }
  auto t1 = std::chrono::steady_clock::now();
  *total_steps = steps;
  *total_additions = adds;
  *checksum = cs;
  return std::chrono::duration<double>(t1 - t0).count();
}

}  // extern "C"
