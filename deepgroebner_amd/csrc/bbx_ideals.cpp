#include "bbx_ideals.h"
#include "bbx_common.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <sstream>

namespace bbx {

// ---------------------------------------------------------------- GF(32003), polynomials.h:14, cpp:11-23
int coef_norm(long long i) { return (int)((i < 0) ? (i % kP) + kP : i % kP); }
int coef_inv(int a) {
  long long r = 1, b = a;
  for (int e = kP - 2; e; e >>= 1) { if (e & 1) r = r * b % kP; b = b * b % kP; }
  return (int)r;
}

bool mono_gt(const HTerm& a, const HTerm& b) {
  if (a.deg != b.deg) return a.deg > b.deg;
  for (int i = kN - 1; i >= 0; i--) {
    if (b.e[i] > a.e[i]) return true;
    if (a.e[i] > b.e[i]) return false;
  }
  return false;
}

static HTerm make_term(int c, const std::array<int, kN>& e) {
  HTerm t; t.c = coef_norm(c); t.e = e; t.deg = 0;
  for (int x : e) t.deg += x;
  return t;
}

HPoly poly_from_terms(std::vector<HTerm> ts) {
  HPoly f;
  // the reference constructor calls std::sort with this comparator (polynomials.cpp:134-135, 142-143)
  std::sort(ts.begin(), ts.end(), [](const HTerm& a, const HTerm& b) { return mono_gt(a, b); });
  f.t = std::move(ts);
  f.sugar = f.t.empty() ? 0 : f.t[0].deg;
  return f;
}

HPoly poly_add(const HPoly& a, const HPoly& b) {
  HPoly g;
  g.sugar = std::max(a.sugar, b.sugar);
  size_t i = 0, j = 0;
  while (i < a.t.size() && j < b.t.size()) {
    if (mono_gt(a.t[i], b.t[j])) g.t.push_back(a.t[i++]);
    else if (mono_gt(b.t[j], a.t[i])) g.t.push_back(b.t[j++]);
    else {
      int c = coef_norm((long long)a.t[i].c + b.t[j].c);
      if (c != 0) { HTerm t = a.t[i]; t.c = c; g.t.push_back(t); }
      i++; j++;
    }
  }
  for (; i < a.t.size(); i++) g.t.push_back(a.t[i]);
  for (; j < b.t.size(); j++) g.t.push_back(b.t[j]);
  return g;
}

// ---------------------------------------------------------------- minimalize / interreduce, buchberger.cpp:102-122
static bool divides(const HTerm& a, const HTerm& b) {   // a | b
  for (int i = 0; i < kN; i++) if (a.e[i] > b.e[i]) return false;
  return true;
}
static HPoly scaled_shift(const HPoly& f, int c, const HTerm& m) {     // Term{c, m} * f  (polynomials.cpp:196-202)
  HPoly g; g.sugar = m.deg + f.sugar;
  for (auto& t : f.t) {
    HTerm u; u.c = coef_norm((long long)c * t.c); u.deg = t.deg + m.deg;
    for (int i = 0; i < kN; i++) u.e[i] = t.e[i] + m.e[i];
    g.t.push_back(u);
  }
  return g;
}
static HPoly negated(const HPoly& f) { HPoly g = f; for (auto& t : g.t) t.c = coef_norm((long long)(kP - 1) * t.c); return g; }

HPoly poly_reduce(const HPoly& g, const std::vector<HPoly>& F) {
  HPoly r, h = g;
  while (!h.t.empty()) {
    bool found = false;
    for (const HPoly& f : F) {
      if (divides(f.t[0], h.t[0])) {
        HTerm q; q.deg = h.t[0].deg - f.t[0].deg;
        for (int i = 0; i < kN; i++) q.e[i] = h.t[0].e[i] - f.t[0].e[i];
        const int c = coef_norm((long long)h.t[0].c * coef_inv(f.t[0].c));
        h = poly_add(h, negated(scaled_shift(f, c, q)));
        found = true;
        break;
      }
    }
    if (!found) {
      HPoly lt; lt.t = {h.t[0]}; lt.sugar = h.t[0].deg;
      r = poly_add(r, lt);
      h = poly_add(h, negated(lt));
    }
  }
  return poly_add(r, h);
}

std::vector<HPoly> minimalize(const std::vector<HPoly>& G) {
  std::vector<HPoly> s = G, out;
  std::sort(s.begin(), s.end(), [](const HPoly& f, const HPoly& g) { return mono_gt(g.t[0], f.t[0]); });   // std::sort, like the reference
  for (const HPoly& g : s) {
    bool ok = true;
    for (const HPoly& f : out) if (divides(f.t[0], g.t[0])) { ok = false; break; }
    if (ok) out.push_back(g);
  }
  return out;
}

std::vector<HPoly> interreduce(const std::vector<HPoly>& G) {
  std::vector<HPoly> out;
  for (const HPoly& g : G) {
    HPoly lt; lt.t = {g.t[0]}; lt.sugar = g.t[0].deg;
    HPoly tail = poly_add(g, negated(lt));
    HPoly s = poly_add(poly_reduce(tail, G), lt);
    HTerm one; one.c = 1; one.deg = 0; one.e = {};
    out.push_back(scaled_shift(s, coef_inv(g.t[0].c), one));
  }
  return out;
}

// ---------------------------------------------------------------- libstdc++ 11 <random>, restated
void MinStd0::seed(long long s) {   // linear_congruential_engine<uint_fast32_t,16807,0,2147483647>::seed
  x = (uint64_t)s % 2147483647ull;
  if (x == 0) x = 1;
}
uint64_t MinStd0::next() { x = x * 16807ull % 2147483647ull; return x; }

static const uint64_t kRngMin = 1, kRngMax = 2147483646ull;

// uniform_int_distribution<int>{a,b}(rng): downscaling branch with two divisions (uniform_int_dist.h)
static int uniform_int(MinStd0& r, int a, int b) {
  const uint64_t urngrange = kRngMax - kRngMin;
  const uint64_t urange = (uint64_t)((long long)b - (long long)a);
  uint64_t ret;
  if (urngrange > urange) {
    const uint64_t uerange = urange + 1;
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    do ret = r.next() - kRngMin; while (ret >= past);
    ret /= scaling;
  } else {
    ret = r.next() - kRngMin;
  }
  return (int)(ret + (uint64_t)(long long)a);
}
// generate_canonical<double,53>: two engine draws for minstd_rand0 (random.tcc)
static double canonical(MinStd0& r) {
  const long double R = (long double)kRngMax - (long double)kRngMin + 1.0L;
  double sum = 0.0, tmp = 1.0;
  for (int k = 2; k != 0; --k) {
    sum += (double)(r.next() - kRngMin) * tmp;
    tmp = (double)((long double)tmp * R);
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = std::nextafter(1.0, 0.0);
  return ret;
}
struct Discrete {               // discrete_distribution<int>::param_type::_M_initialize + operator()
  std::vector<double> prob, cp;
  void init(const std::vector<int>& w) {
    prob.clear(); cp.clear();
    if (w.size() < 2) return;
    double sum = 0.0;
    for (int x : w) sum += (double)x;
    for (int x : w) prob.push_back((double)x / sum);
    double acc = 0.0;
    for (double p : prob) { acc += p; cp.push_back(acc); }
    cp.back() = 1.0;
  }
  int draw(MinStd0& r) const {
    if (cp.empty()) return 0;
    double p = canonical(r);
    return (int)(std::lower_bound(cp.begin(), cp.end(), p) - cp.begin());
  }
};
// std::poisson_distribution<int>{mean} of libstdc++ 11 (bits/random.tcc:1261-1404) over minstd_rand0: products of
// canonical draws below mean 12; from 12 on Devroye's rejection algorithm (Non-Uniform Random Variate Generation, X.3.3-4
// + errata) with the distribution's own std::normal_distribution<double> (Marsaglia polar method, which keeps the second
// variate of a pair for the next call: that state belongs to the generator and travels with its copies, as the
// reference's member `length_dist` does, ideals.h:227).  Same libm calls as the library, so the same doubles.
struct Poisson {
  double mean = 0, lm_thr = 0, lfm = 0, sm = 0, d = 0, scx = 0, cx1 = 0, c2b = 0, cb = 0;
  bool saved_available = false; double saved = 0;
  void init(double m_) {
    mean = m_;
    if (mean >= 12) {
      const double m = std::floor(mean);
      lm_thr = std::log(mean); lfm = std::lgamma(m + 1); sm = std::sqrt(m);
      const double pi_4 = 0.7853981633974483096156608458198757L;
      const double dx = std::sqrt(2 * m * std::log(32 * m / pi_4));
      d = std::round(std::max<double>(6.0, std::min(m, dx)));
      const double cx = 2 * m + d;
      scx = std::sqrt(cx / 2); cx1 = 1 / cx;
      c2b = std::sqrt(pi_4 * cx) * std::exp(cx1);
      cb = 2 * cx * std::exp(-d * cx1 * (1 + d / 2)) / d;
    } else lm_thr = std::exp(-mean);
  }
  double normal(MinStd0& r) {                  // normal_distribution<double>(0, 1), random.tcc:1802-1835
    double ret;
    if (saved_available) { saved_available = false; ret = saved; }
    else {
      double x, y, r2;
      do { x = 2.0 * canonical(r) - 1.0; y = 2.0 * canonical(r) - 1.0; r2 = x * x + y * y; } while (r2 > 1.0 || r2 == 0.0);
      const double mult = std::sqrt(-2 * std::log(r2) / r2);
      saved = x * mult; saved_available = true;
      ret = y * mult;
    }
    return ret * 1.0 + 0.0;
  }
  int draw(MinStd0& r) {
    if (mean < 12) {
      int x = 0;
      double prod = 1.0;
      do { prod *= canonical(r); x += 1; } while (prod > lm_thr);
      return x - 1;
    }
    double x;
    const double naf = (1 - std::numeric_limits<double>::epsilon()) / 2;
    const double thr = std::numeric_limits<int>::max() + naf;
    const double m = std::floor(mean);
    const double spi_2 = 1.2533141373155002512078826424055226L;
    const double c1 = sm * spi_2, c2 = c2b + c1, c3 = c2 + 1, c4 = c3 + 1;
    const double k178 = 0.0128205128205128205128205128205128L, e178 = 1.0129030479320018583185514777512983L;
    const double c5 = c4 + e178, c = cb + c5, cx2 = 2 * (2 * m + d);
    bool reject = true;
    do {
      const double u = c * canonical(r);
      const double e = -std::log(1.0 - canonical(r));
      double w = 0.0;
      if (u <= c1) {
        const double n = normal(r);
        const double y = -std::abs(n) * sm - 1;
        x = std::floor(y);
        w = -n * n / 2;
        if (x < -m) continue;
      } else if (u <= c2) {
        const double n = normal(r);
        const double y = 1 + std::abs(n) * scx;
        x = std::ceil(y);
        w = y * (2 - y) * cx1;
        if (x > d) continue;
      } else if (u <= c3) x = -1;
      else if (u <= c4) x = 0;
      else if (u <= c5) { x = 1; w = k178; }
      else {
        const double v = -std::log(1.0 - canonical(r));
        const double y = d + v * cx2 / d;
        x = std::ceil(y);
        w = -d * cx1 * (1 + y / 2);
      }
      reject = (w - e - x * lm_thr > lfm - std::lgamma(x + m + 1));
      reject |= x + m >= thr;
    } while (reject);
    return (int)(x + m + naf);
  }
};

// ---------------------------------------------------------------- ideals.cpp
std::vector<HPoly> cyclic(int n) {
  std::vector<HPoly> F;
  for (int d = 1; d < n; d++) {
    std::vector<HTerm> p;
    for (int i = 0; i < n; i++) {
      std::array<int, kN> e{};
      for (int k = 0; k < d; k++) e[(i + k) % n] = 1;
      p.push_back(make_term(1, e));
    }
    F.push_back(poly_from_terms(p));
  }
  std::array<int, kN> e{};
  for (int i = 0; i < n; i++) e[i] = 1;
  F.push_back(poly_from_terms({make_term(1, e), make_term(-1, std::array<int, kN>{})}));
  return F;
}

std::vector<std::array<int, kN>> basis(int n, int d) {
  std::vector<int> a;
  for (int i = 0; i < d; i++) a.push_back(0);
  for (int i = 0; i < n - 1; i++) a.push_back(1);
  std::vector<std::array<int, kN>> B;
  do {                                   // stars (0) and bars (1); std::next_permutation order matters
    std::array<int, kN> e{};
    int index = 0;
    for (int v : a) { if (v == 0) e[index]++; else index++; }
    B.push_back(e);
  } while (std::next_permutation(a.begin(), a.end()));
  return B;
}

static int binomial(int n, int k) { return (k == 0 || k == n) ? 1 : binomial(n - 1, k - 1) + binomial(n - 1, k); }

static std::vector<int> degree_weights(int n, int d, DistType dist, bool constants) {  // ideals.cpp:75-100
  std::vector<int> count;
  count.push_back(constants ? 1 : 0);
  switch (dist) {
    case DistType::Uniform: for (int i = 1; i < d + 1; i++) count.push_back(binomial(n + i - 1, n - 1)); break;
    case DistType::Weighted: for (int i = 0; i < d; i++) count.push_back(1); break;
    case DistType::Maximum: for (int i = 0; i < d - 1; i++) count.push_back(0); count.push_back(1); break;
  }
  return count;
}
std::vector<double> degree_probabilities(int n, int d, DistType dist, bool constants) {
  Discrete dd; dd.init(degree_weights(n, d, dist, constants));
  return dd.prob;
}

namespace {

using Bases = std::vector<std::vector<std::array<int, kN>>>;

class FixedGen : public IdealGen {
 public:
  explicit FixedGen(const HIdeal& F) : F_(F) {
    n_ = 0;                               // ideals.cpp:146-154: the max variable INDEX (sic)
    for (auto& f : F_) for (auto& t : f.t) for (int i = 0; i < kN; i++) if (t.e[i] != 0) n_ = std::max(n_, i);
  }
  bool next(HIdeal& out, std::string*) override { out = F_; return true; }
  int nvars() const override { return n_; }
  std::unique_ptr<IdealGen> clone() const override { return std::make_unique<FixedGen>(*this); }
  bool fixed() const override { return true; }
  int max_terms_hint() const override { size_t m = 1; for (auto& f : F_) m = std::max(m, f.t.size()); return (int)m; }
  int npolys() const override { return (int)F_.size(); }
 private:
  HIdeal F_;
  int n_;
};

class ListGen : public IdealGen {
 public:
  ListGen(std::shared_ptr<const std::vector<HIdeal>> ideals, int first, int stride, int nvars)
      : ideals_(std::move(ideals)), at_(first), stride_(stride), n_(nvars) {
    maxt_ = 1; maxp_ = 1;
    for (auto& F : *ideals_) { maxp_ = std::max(maxp_, (int)F.size()); for (auto& f : F) maxt_ = std::max(maxt_, (int)f.t.size()); }
  }
  bool next(HIdeal& out, std::string*) override {
    const size_t n = ideals_->size();
    out = (*ideals_)[(size_t)at_ % n];
    at_ = (int)(((size_t)at_ + (size_t)stride_) % (n * (size_t)std::max(stride_, 1)));
    return true;
  }
  int nvars() const override { return n_; }
  std::unique_ptr<IdealGen> clone() const override { return std::make_unique<ListGen>(*this); }
  int max_terms_hint() const override { return maxt_; }
  int npolys() const override { return maxp_; }
  void set_first(int first) { at_ = first; }
 private:
  std::shared_ptr<const std::vector<HIdeal>> ideals_;
  int at_, stride_, n_, maxt_, maxp_;
};

class RandomBase : public IdealGen {
 public:
  RandomBase(int n, int d, int s, DistType dist, bool constants, bool homogeneous)
      : n_(n), s_(s), homogeneous_(homogeneous) {
    auto b = std::make_shared<Bases>();
    for (int i = 0; i < d + 1; i++) b->push_back(basis(n, i));
    bases_ = b;
    degree_.init(degree_weights(n, d, dist, constants));
    rng_.seed(5489);                      // the reference seeds from std::random_device; callers seed explicitly
  }
  void seed(long long s) override { rng_.seed(s); }
  int nvars() const override { return n_; }
  int npolys() const override { return s_; }
 protected:
  bool base_table(int W, std::vector<uint32_t>* out, uint32_t kind_flags, double lm_thr) const {
    const int d = (int)bases_->size() - 1, slots = 2 * W;
    if (d > BBX_GEN_MAXDEG || n_ > slots - 1 || degree_.cp.size() > 64) return false;
    size_t total = 0;
    for (auto& B : *bases_) total += B.size();
    if (total > (size_t)1 << 22) return false;
    std::vector<uint32_t>& t = *out;
    t.assign(BBX_GEN_MONO + total * W, 0u);
    t[0] = (uint32_t)n_; t[1] = (uint32_t)d; t[2] = (uint32_t)s_; t[3] = (homogeneous_ ? 1u : 0u) | kind_flags;
    memcpy(&t[6], &lm_thr, 8);
    t[4] = (uint32_t)degree_.cp.size(); t[5] = (uint32_t)W;
    for (size_t i = 0; i < 64; i++) {
      const double v = i < degree_.cp.size() ? degree_.cp[i] : HUGE_VAL;
      memcpy(&t[BBX_GEN_CP + 2 * i], &v, 8);
    }
    size_t at = 0;
    for (int i = 0; i <= d; i++) {
      const auto& B = (*bases_)[i];
      // uniform_int_distribution<int>(0, len - 1) on minstd_rand0: scaling = (2^31 - 3) / len, past = len * scaling
      const uint64_t urngrange = kRngMax - kRngMin, len = B.size();
      const uint64_t scaling = urngrange / len, past = len * scaling;
      t[BBX_GEN_DEG + 8 * i] = (uint32_t)at; t[BBX_GEN_DEG + 8 * i + 1] = (uint32_t)len;
      t[BBX_GEN_DEG + 8 * i + 2] = (uint32_t)scaling; t[BBX_GEN_DEG + 8 * i + 3] = (uint32_t)past;
      t[BBX_GEN_DEG + 8 * i + 4] = (uint32_t)((1ull << 32) / scaling);
      for (auto& e : B) {
        uint32_t s[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int v = 0; v < n_; v++) s[v] = (uint32_t)e[v];
        s[slots - 1] = (uint32_t)i;
        for (int w = 0; w < W; w++) t[BBX_GEN_MONO + at * W + w] = s[2 * w] | (s[2 * w + 1] << 16);
        at++;
      }
    }
    return true;
  }
  HTerm choice(int d, int c) {            // choice(): fresh uniform_int_distribution(0, len-1), ideals.h:68-73
    const auto& B = (*bases_)[d];
    return make_term(c, B[uniform_int(rng_, 0, (int)B.size() - 1)]);
  }
  int n_, s_;
  bool homogeneous_;
  std::shared_ptr<const Bases> bases_;
  Discrete degree_;
  MinStd0 rng_;
};

class BinomialGen : public RandomBase {   // ideals.cpp:156-201
 public:
  BinomialGen(int n, int d, int s, DistType dist, bool constants, bool homogeneous, bool pure)
      : RandomBase(n, d, s, dist, constants, homogeneous), pure_(pure) {}
  bool next(HIdeal& F, std::string* err) override {
    F.clear();
    for (int i = 0; i < s_; i++) {
      int c = pure_ ? coef_norm(-1) : uniform_int(rng_, 1, kP - 1);
      int d1, d2;
      if (homogeneous_) d1 = d2 = degree_.draw(rng_);
      else { d1 = degree_.draw(rng_); d2 = degree_.draw(rng_); }
      bool success = false;
      for (int trials = 0; trials < 1000; trials++) {
        HTerm m1 = choice(d1, 1), m2 = choice(d2, 1);
        HPoly f;
        if (mono_gt(m2, m1)) { m1.c = c; f.t = {m2, m1}; }
        else if (mono_gt(m1, m2)) { m2.c = c; f.t = {m1, m2}; }
        else continue;
        f.sugar = f.t[0].deg;
        F.push_back(f);
        success = true;
        break;
      }
      if (!success) { if (err) *err = "failed to generate two distinct random monomials after 1000 trials"; return false; }
    }
    return true;
  }
  std::unique_ptr<IdealGen> clone() const override { return std::make_unique<BinomialGen>(*this); }
  int max_terms_hint() const override { return 2; }
  bool device_table(int W, std::vector<uint32_t>* out) const override { return base_table(W, out, pure_ ? 2u : 0u, 0.0); }
 private:
  bool pure_;
};

class RandomGen : public RandomBase {     // ideals.cpp:203-231
 public:
  RandomGen(int n, int d, int s, double lam, DistType dist, bool constants, bool homogeneous)
      : RandomBase(n, d, s, dist, constants, homogeneous), lam_(lam), lm_thr_(std::exp(-lam)) { length_.init(lam); }
  bool next(HIdeal& F, std::string* err) override {
    F.clear();
    for (int i = 0; i < s_; i++) {
      HPoly f;
      int terms = 2 + length_.draw(rng_);
      int d = degree_.draw(rng_);
      for (int j = 0; j < terms; j++) {
        int c = uniform_int(rng_, 1, kP - 1);
        HTerm t = choice(d, c);
        HPoly single; single.t = {t}; single.sugar = t.deg;
        f = poly_add(f, single);
        if (!homogeneous_) d = degree_.draw(rng_);
      }
      if (f.t.empty()) { if (err) *err = "random polynomial cancelled to zero (undefined in the reference)"; return false; }
      int inv = coef_inv(f.t[0].c);       // Term{1 / f.LC(), {}} * f
      for (auto& t : f.t) t.c = coef_norm((long long)t.c * inv);
      F.push_back(f);
    }
    return true;
  }
  std::unique_ptr<IdealGen> clone() const override { return std::make_unique<RandomGen>(*this); }
  // 2 + Poisson(lam) terms: the queue slots are sized for the mean + 10 standard deviations (a longer draw is reported
  // as a capacity error, not truncated)
  int max_terms_hint() const override { return lam_ < 12.0 ? 64 : (int)(18 + lam_ + 10 * std::sqrt(lam_)); }
  // means >= 12 are drawn on the host only: the rejection sampler's log / lgamma / exp would have to give the host
  // library's doubles bit for bit on the device
  bool device_table(int W, std::vector<uint32_t>* out) const override { return lam_ < 12.0 && base_table(W, out, 4u, lm_thr_); }
 private:
  double lam_, lm_thr_;
  Poisson length_;
};

}  // namespace

std::unique_ptr<IdealGen> make_fixed(const HIdeal& F) { return std::make_unique<FixedGen>(F); }
std::unique_ptr<IdealGen> make_list(std::shared_ptr<const std::vector<HIdeal>> ideals, int first, int stride, int nvars) {
  return std::make_unique<ListGen>(std::move(ideals), first, stride, nvars);
}

std::unique_ptr<IdealGen> parse_ideal_dist(const std::string& ideal_dist, std::string* err) {
  std::vector<std::string> a;
  std::string arg;
  std::istringstream iss(ideal_dist);
  while (std::getline(iss, arg, '-')) a.push_back(arg);
  auto dist_of = [](const std::string& s, DistType* out) {
    if (s == "uniform") { *out = DistType::Uniform; return true; }
    if (s == "weighted") { *out = DistType::Weighted; return true; }
    if (s == "maximum") { *out = DistType::Maximum; return true; }
    return false;
  };
  auto has = [&a](const char* s) { return std::find(a.begin(), a.end(), s) != a.end(); };
  auto fail = [&](const char* m) { if (err) *err = std::string(m) + ": '" + ideal_dist + "'"; return std::unique_ptr<IdealGen>(); };
  try {
    if (a.size() >= 2 && a[0] == "cyclic") {
      int n = std::stoi(a[1]);
      if (n < 2 || n > kN) return fail("cyclic-n needs 2 <= n <= 8");
      return make_fixed(cyclic(n));
    }
    if (a.size() < 4) return fail("unrecognised ideal distribution");
    int n = std::stoi(a[0]), d = std::stoi(a[1]), s = std::stoi(a[2]);
    if (n < 1 || n > kN || d < 1 || s < 1) return fail("bad n-d-s in ideal distribution");
    DistType dt;
    if (dist_of(a[3], &dt)) return std::make_unique<BinomialGen>(n, d, s, dt, has("consts"), has("homog"), has("pure"));
    if (a.size() < 5) return fail("unrecognised ideal distribution");
    double lam = std::stod(a[3]);
    if (!dist_of(a[4], &dt)) dt = DistType::Uniform;   // dist_types[unknown] default-inserts Uniform
    return std::make_unique<RandomGen>(n, d, s, lam, dt, has("consts"), has("homog"));
  } catch (const std::exception&) {
    return fail("unparsable ideal distribution");
  }
}

// ---------------------------------------------------------------- text format, polynomials.cpp:226-300
// The reference parses by mutual recursion over an istringstream; the language it accepts is
//   poly := term*            term := ('+' | '-')* ( INT ['*' mono] | mono )
//   mono := <end> | VAR ['^' INT] ['*' mono]          VAR := 'a'..'h'
// and the value is the sum of the terms (like monomials merge, zero sums vanish).  Written here as one left-to-right
// scan.  Where the reference runs into undefined behaviour this parser reports an error instead: variable 'i'
// (index N, polynomials.cpp:228), negative or missing exponents, coefficients that are 0 mod 32003 (the reference
// keeps such a term in the polynomial) and empty polynomials inside an ideal.
namespace {
struct Scan {
  const std::string& s; size_t at = 0;
  int peek() const { return at < s.size() ? (unsigned char)s[at] : -1; }
  int get() { return at < s.size() ? (unsigned char)s[at++] : -1; }
  bool number(long long* v) {           // digits only
    if (peek() < '0' || peek() > '9') return false;
    long long x = 0;
    while (peek() >= '0' && peek() <= '9') { x = x * 10 + (get() - '0'); if (x > 2147483647LL) return false; }
    *v = x; return true;
  }
};
}  // namespace

bool parse_polynomial(const std::string& text, HPoly& out, std::string* err) {
  auto fail = [&](const std::string& m, size_t at) { if (err) *err = m + " at column " + std::to_string(at + 1) + " of '" + text + "'"; return false; };
  Scan sc{text};
  std::vector<HTerm> ts;
  while (sc.peek() != -1) {
    long long sign = 1;
    while (sc.peek() == '+' || sc.peek() == '-') if (sc.get() == '-') sign = -sign;    // Term{-1,{}} * parse_term
    HTerm t; t.c = 1; t.e = {}; t.deg = 0;
    bool mono = true;
    if (sc.peek() >= '0' && sc.peek() <= '9') {
      long long c;
      if (!sc.number(&c)) return fail("coefficient out of range", sc.at);
      t.c = coef_norm(c);
      if (sc.peek() == '*') sc.get(); else mono = false;      // "3" is a constant; "3a" is 3 + a
    }
    while (mono && sc.peek() != -1) {                          // parse_monomial; "3*" at the end is the constant 3
      const size_t vat = sc.at;
      const int v = sc.get() - 'a';
      if (v < 0 || v >= kN) return fail("invalid variable name", vat);
      long long pw = 1;
      if (sc.peek() == '^') {
        sc.get();
        if (!sc.number(&pw)) return fail("bad exponent", sc.at);
      }
      t.e[v] += (int)pw; t.deg += (int)pw;
      if (t.deg > 65535) return fail("degree above 65535", vat);
      if (sc.peek() == '*') sc.get(); else mono = false;
    }
    t.c = coef_norm(sign * t.c);
    if (t.c == 0) return fail("coefficient is zero in GF(32003)", sc.at ? sc.at - 1 : 0);
    ts.push_back(t);
  }
  // Polynomial{t} + (Polynomial{t'} + ...): a sum of single terms, evaluated from the right
  HPoly acc;
  for (size_t k = ts.size(); k-- > 0;) {
    HPoly one; one.t = {ts[k]}; one.sugar = ts[k].deg;
    acc = poly_add(one, acc);
  }
  out = acc;
  return true;
}

bool parse_ideal_string(const std::string& text, HIdeal& out, std::string* err) {
  out.clear();
  std::istringstream iss(text);
  std::string piece;
  while (std::getline(iss, piece, '|')) {
    while (!piece.empty() && (piece.back() == '\r' || piece.back() == '\n')) piece.pop_back();
    HPoly f;
    if (!parse_polynomial(piece, f, err)) return false;
    if (f.t.empty()) { if (err) *err = "zero polynomial in ideal '" + text + "'"; return false; }
    out.push_back(std::move(f));
  }
  if (out.empty()) { if (err) *err = "empty ideal string"; return false; }
  return true;
}

std::string format_polynomial(const HPoly& f) {
  if (f.t.empty()) return "0";
  std::string s;
  for (size_t k = 0; k < f.t.size(); k++) {
    const HTerm& t = f.t[k];
    const int c = t.c > kP / 2 ? t.c - kP : t.c;            // signed representative
    const int a = c < 0 ? -c : c;
    if (c < 0) s += '-'; else if (k) s += '+';
    if (t.deg == 0) { s += std::to_string(a); continue; }
    if (a != 1) { s += std::to_string(a); s += '*'; }
    bool first = true;
    for (int v = 0; v < kN; v++) {
      if (!t.e[v]) continue;
      if (!first) s += '*';
      first = false;
      s += (char)('a' + v);
      if (t.e[v] != 1) { s += '^'; s += std::to_string(t.e[v]); }
    }
  }
  return s;
}

}  // namespace bbx
