// Host side of libbbx: the ideal-distribution generators (reference deepgroebner/ideals.{h,cpp}).
// Seeded streams are bit-identical to the reference built with libstdc++ 11: the engine
// (minstd_rand0) and the three distributions it uses are implemented here from the published
// libstdc++ algorithms instead of calling <random>, so the streams do not depend on the C++
// runtime this library happens to be built against.
#pragma once
#include <array>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace bbx {

constexpr int kP = 32003;   // polynomials.h:10
constexpr int kN = 8;       // polynomials.h:29

struct HTerm { int c; std::array<int, kN> e; int deg; };
struct HPoly { std::vector<HTerm> t; int sugar = 0; };
using HIdeal = std::vector<HPoly>;

int coef_norm(long long i);
int coef_inv(int a);
bool mono_gt(const HTerm& a, const HTerm& b);   // grevlex on the exponent vectors, polynomials.cpp:60-74
HPoly poly_from_terms(std::vector<HTerm> ts);   // Polynomial ctor: sort descending, sugar = deg LT
HPoly poly_add(const HPoly& a, const HPoly& b); // polynomials.cpp:148-177

// host-side Groebner-basis post-processing (buchberger.cpp:102-122), used by bbx_reduced_basis
HPoly poly_reduce(const HPoly& g, const std::vector<HPoly>& F);          // buchberger.cpp:24-49 (remainder only)
std::vector<HPoly> minimalize(const std::vector<HPoly>& G);              // buchberger.cpp:102-112
std::vector<HPoly> interreduce(const std::vector<HPoly>& G);             // buchberger.cpp:115-122

class MinStd0 {               // std::default_random_engine
 public:
  void seed(long long s);
  uint64_t next();
  uint64_t x = 1;
};

enum class DistType { Uniform = 0, Weighted = 1, Maximum = 2 };

class IdealGen {
 public:
  virtual ~IdealGen() {}
  virtual bool next(HIdeal& out, std::string* err) = 0;  // false = the reference would have thrown
  virtual void seed(long long) {}
  virtual int nvars() const = 0;
  virtual std::unique_ptr<IdealGen> clone() const = 0;
  virtual bool fixed() const { return false; }
  virtual int max_terms_hint() const = 0;                // upper bound on terms per generator (slot sizing)
  virtual int npolys() const = 0;
  // the table a kernel needs to draw this generator's ideals itself (layout: BBX_GEN_* in bbx_common.h), W = words per
  // packed monomial; false: this generator only runs on the host
  virtual bool device_table(int W, std::vector<uint32_t>* out) const { (void)W; (void)out; return false; }
};

std::vector<HPoly> cyclic(int n);                         // ideals.cpp:16-36
std::vector<std::array<int, kN>> basis(int n, int d);     // ideals.cpp:39-64
std::vector<double> degree_probabilities(int n, int d, DistType dist, bool constants);
std::unique_ptr<IdealGen> make_fixed(const HIdeal& F);
// a generator that walks a shared list of ideals: first, first + stride, ... (wrapping around)
std::unique_ptr<IdealGen> make_list(std::shared_ptr<const std::vector<HIdeal>> ideals, int first, int stride, int nvars);
std::unique_ptr<IdealGen> parse_ideal_dist(const std::string& dist, std::string* err);  // ideals.cpp:103-143

// text format of data/stats/<dist>/<dist>.csv: polynomials "413*a^2*b^5*c+32*d^2-5" joined by '|'
// (parse_polynomial polynomials.cpp:226-300, parse_ideal_string scripts/make_strat.cpp:12-19)
bool parse_polynomial(const std::string& text, HPoly& out, std::string* err);
bool parse_ideal_string(const std::string& text, HIdeal& out, std::string* err);
std::string format_polynomial(const HPoly& f);   // the same format, signed coefficient representatives like Macaulay2 prints

}  // namespace bbx
