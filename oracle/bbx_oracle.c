/* TEST INFRASTRUCTURE — the parity oracle (see bbx_oracle.h).  NOT product code.
 *
 * Plain-C restatement of the reference's algorithm for the BuchbergerEnv step
 * path.  Every function cites the reference file:line it follows
 * (paths relative to /root/reference/deepgroebner/).  Where the reference leans
 * on libstdc++ (std::sort, <random>), the published libstdc++ 11 algorithm is
 * restated (GCC 11.4 is what the reference's seeded known answers were produced
 * with: tests/test_ideals.cpp:123-144 pass under it).
 *
 * Single-threaded, scalar, obviously-correct-over-fast.
 */
#define _POSIX_C_SOURCE 200809L
#include "bbx_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define BO_P 32003 /* polynomials.h:10 */
#define BO_N 8     /* polynomials.h:29 */

/* ======================================================================== */
/* Coefficient — GF(32003)                          polynomials.h:10-26      */
/* ======================================================================== */

/* polynomials.h:14  Coefficient(int i): c{(i < 0) ? (i % P) + P : i % P} */
static int coef_norm(long long i) { return (int)((i < 0) ? (i % BO_P) + BO_P : i % BO_P); }
static int coef_add(int a, int b) { return coef_norm((long long)a + b); } /* h:16 */
static int coef_sub(int a, int b) { return coef_norm((long long)a - b); } /* h:17 */
static int coef_mul(int a, int b) { return coef_norm((long long)a * b); } /* h:18 */
/* polynomials.cpp:11-23: inverse of c2 by the extended Euclidean algorithm,
 * result c1 * a re-normalised by the implicit constructor */
static int coef_div(int c1, int c2) {
  int a = 0, a_ = 1;
  int b = BO_P, b_ = c2;
  while (b_ != 0) {
    int q = b / b_;
    int t = a - q * a_; a = a_; a_ = t;
    t = b - q * b_; b = b_; b_ = t;
  }
  return coef_norm((long long)c1 * a);
}

/* ======================================================================== */
/* Monomial                                        polynomials.h:29-55       */
/* ======================================================================== */
typedef struct { int e[BO_N]; int deg; } mono;

static mono mono_make(const int* e) { /* cpp:34-38 */
  mono m; m.deg = 0;
  for (int i = 0; i < BO_N; i++) { m.e[i] = e[i]; m.deg += e[i]; }
  return m;
}
static mono mono_one(void) { mono m; memset(&m, 0, sizeof m); return m; }
static mono mono_mul(const mono* a, const mono* b) { /* cpp:41-47 */
  mono m; m.deg = a->deg + b->deg;
  for (int i = 0; i < BO_N; i++) m.e[i] = a->e[i] + b->e[i];
  return m;
}
static mono mono_div(const mono* a, const mono* b) { /* cpp:50-57 (no underflow check) */
  mono m; m.deg = a->deg - b->deg;
  for (int i = 0; i < BO_N; i++) m.e[i] = a->e[i] - b->e[i];
  return m;
}
/* cpp:60-74 grevlex: higher degree wins; else scanning from the last variable
 * the SMALLER exponent wins */
static int mono_gt(const mono* a, const mono* b) {
  if (a->deg > b->deg) return 1;
  if (b->deg > a->deg) return 0;
  for (int i = BO_N - 1; i >= 0; i--) {
    if (b->e[i] > a->e[i]) return 1;
    if (a->e[i] > b->e[i]) return 0;
  }
  return 0;
}
static int mono_lt(const mono* a, const mono* b) { return mono_gt(b, a); } /* h:46 */
static int mono_eq(const mono* a, const mono* b) { /* cpp:77-81 (ignores degree) */
  for (int i = 0; i < BO_N; i++) if (a->e[i] != b->e[i]) return 0;
  return 1;
}
static int mono_divisible(const mono* a, const mono* b) { /* cpp:93-98: b | a */
  for (int i = 0; i < BO_N; i++) if (a->e[i] < b->e[i]) return 0;
  return 1;
}
static mono mono_lcm(const mono* a, const mono* b) { /* cpp:111-118 */
  mono m; m.deg = 0;
  for (int i = 0; i < BO_N; i++) { m.e[i] = a->e[i] > b->e[i] ? a->e[i] : b->e[i]; m.deg += m.e[i]; }
  return m;
}

/* ======================================================================== */
/* libstdc++ std::sort restated (bits/stl_algo.h, GCC 11): introsort with a  */
/* 16-element threshold + final insertion sort, on an int index array.       */
/* cmp(ctx, a, b) is the strict-weak "a before b".                           */
/* ======================================================================== */
typedef int (*cmp_fn)(void* ctx, int a, int b);

static void ss_unguarded_linear_insert(int* v, int last, cmp_fn cmp, void* ctx) {
  int val = v[last];
  int next = last - 1;
  while (cmp(ctx, val, v[next])) { v[last] = v[next]; last = next; next--; }
  v[last] = val;
}
static void ss_insertion_sort(int* v, int first, int last, cmp_fn cmp, void* ctx) {
  if (first == last) return;
  for (int i = first + 1; i != last; i++) {
    if (cmp(ctx, v[i], v[first])) {
      int val = v[i];
      memmove(v + first + 1, v + first, (size_t)(i - first) * sizeof(int));
      v[first] = val;
    } else {
      ss_unguarded_linear_insert(v, i, cmp, ctx);
    }
  }
}
static void ss_push_heap(int* v, int first, int hole, int top, int value, cmp_fn cmp, void* ctx) {
  int parent = (hole - 1) / 2;
  while (hole > top && cmp(ctx, v[first + parent], value)) {
    v[first + hole] = v[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  v[first + hole] = value;
}
static void ss_adjust_heap(int* v, int first, int hole, int len, int value, cmp_fn cmp, void* ctx) {
  int top = hole, child = hole;
  while (child < (len - 1) / 2) {
    child = 2 * (child + 1);
    if (cmp(ctx, v[first + child], v[first + child - 1])) child--;
    v[first + hole] = v[first + child];
    hole = child;
  }
  if ((len & 1) == 0 && child == (len - 2) / 2) {
    child = 2 * (child + 1);
    v[first + hole] = v[first + child - 1];
    hole = child - 1;
  }
  ss_push_heap(v, first, hole, top, value, cmp, ctx);
}
static void ss_heapsort(int* v, int first, int last, cmp_fn cmp, void* ctx) { /* __partial_sort(f,l,l) */
  int len = last - first;
  if (len >= 2) {
    for (int parent = (len - 2) / 2;; parent--) {
      ss_adjust_heap(v, first, parent, len, v[first + parent], cmp, ctx);
      if (parent == 0) break;
    }
  }
  while (last - first > 1) {
    --last;
    int value = v[last];
    v[last] = v[first];
    ss_adjust_heap(v, first, 0, last - first, value, cmp, ctx);
  }
}
static void ss_swap(int* v, int a, int b) { int t = v[a]; v[a] = v[b]; v[b] = t; }
static void ss_move_median_to_first(int* v, int result, int a, int b, int c, cmp_fn cmp, void* ctx) {
  if (cmp(ctx, v[a], v[b])) {
    if (cmp(ctx, v[b], v[c])) ss_swap(v, result, b);
    else if (cmp(ctx, v[a], v[c])) ss_swap(v, result, c);
    else ss_swap(v, result, a);
  } else if (cmp(ctx, v[a], v[c])) ss_swap(v, result, a);
  else if (cmp(ctx, v[b], v[c])) ss_swap(v, result, c);
  else ss_swap(v, result, b);
}
static int ss_unguarded_partition(int* v, int first, int last, int pivot, cmp_fn cmp, void* ctx) {
  for (;;) {
    while (cmp(ctx, v[first], v[pivot])) first++;
    --last;
    while (cmp(ctx, v[pivot], v[last])) --last;
    if (!(first < last)) return first;
    ss_swap(v, first, last);
    first++;
  }
}
static void ss_introsort_loop(int* v, int first, int last, int depth, cmp_fn cmp, void* ctx) {
  while (last - first > 16) {
    if (depth == 0) { ss_heapsort(v, first, last, cmp, ctx); return; }
    --depth;
    int mid = first + (last - first) / 2;
    ss_move_median_to_first(v, first, first + 1, mid, last - 1, cmp, ctx);
    int cut = ss_unguarded_partition(v, first + 1, last, first, cmp, ctx);
    ss_introsort_loop(v, cut, last, depth, cmp, ctx);
    last = cut;
  }
}
static void std_sort(int* v, int n, cmp_fn cmp, void* ctx) {
  if (n <= 0) return;
  int lg = 0;
  for (int t = n; t > 1; t >>= 1) lg++;
  ss_introsort_loop(v, 0, n, 2 * lg, cmp, ctx);
  if (n > 16) {
    ss_insertion_sort(v, 0, 16, cmp, ctx);
    for (int i = 16; i != n; i++) ss_unguarded_linear_insert(v, i, cmp, ctx);
  } else {
    ss_insertion_sort(v, 0, n, cmp, ctx);
  }
}

/* ======================================================================== */
/* Term / Polynomial                               polynomials.h:58-94       */
/* ======================================================================== */
typedef struct { int c; mono m; } term;
typedef struct { term* t; int n, cap; int sug; } poly;

static void poly_init(poly* f) { f->t = NULL; f->n = 0; f->cap = 0; f->sug = 0; }
static void poly_free(poly* f) { free(f->t); poly_init(f); }
static void poly_push(poly* f, term t) {
  if (f->n == f->cap) { f->cap = f->cap ? 2 * f->cap : 4; f->t = (term*)realloc(f->t, (size_t)f->cap * sizeof(term)); }
  f->t[f->n++] = t;
}
static poly poly_clone(const poly* f) {
  poly g; poly_init(&g);
  if (f->n) { g.t = (term*)malloc((size_t)f->n * sizeof(term)); memcpy(g.t, f->t, (size_t)f->n * sizeof(term)); }
  g.n = g.cap = f->n; g.sug = f->sug;
  return g;
}
static int term_desc_cmp(void* ctx, int a, int b) { /* cpp:134-135 comparator */
  term* t = (term*)ctx;
  return mono_gt(&t[a].m, &t[b].m);
}
/* cpp:131-145: constructor sorts terms descending, sugar = deg(LT) */
static poly poly_from_terms(const term* ts, int n) {
  poly f; poly_init(&f);
  if (n == 0) return f;
  int* idx = (int*)malloc((size_t)n * sizeof(int));
  for (int i = 0; i < n; i++) idx[i] = i;
  std_sort(idx, n, term_desc_cmp, (void*)ts);
  for (int i = 0; i < n; i++) poly_push(&f, ts[idx[i]]);
  free(idx);
  f.sug = f.t[0].m.deg;
  return f;
}
static poly poly_single(term t) { return poly_from_terms(&t, 1); }
/* cpp:148-177 two-pointer merge, zero sums dropped, sugar = max */
static poly poly_add(const poly* f1, const poly* f2) {
  poly g; poly_init(&g);
  g.sug = f1->sug > f2->sug ? f1->sug : f2->sug;
  int i = 0, j = 0;
  while (i < f1->n && j < f2->n) {
    const term* t1 = &f1->t[i]; const term* t2 = &f2->t[j];
    if (mono_gt(&t1->m, &t2->m)) { poly_push(&g, *t1); i++; }
    else if (mono_gt(&t2->m, &t1->m)) { poly_push(&g, *t2); j++; }
    else {
      int c = coef_add(t1->c, t2->c);
      if (c != 0) { term t = *t1; t.c = c; poly_push(&g, t); }
      i++; j++;
    }
  }
  for (; i < f1->n; i++) poly_push(&g, f1->t[i]);
  for (; j < f2->n; j++) poly_push(&g, f2->t[j]);
  return g;
}
/* cpp:180-185: copy, (-1)*coeff each, add */
static poly poly_sub(const poly* f1, const poly* f2) {
  poly f = poly_clone(f2);
  for (int k = 0; k < f.n; k++) f.t[k].c = coef_mul(coef_norm(-1), f.t[k].c);
  poly g = poly_add(f1, &f);
  poly_free(&f);
  return g;
}
static term term_mul(const term* a, const term* b) { term t; t.c = coef_mul(a->c, b->c); t.m = mono_mul(&a->m, &b->m); return t; } /* h:60 */
static term term_div(const term* a, const term* b) { term t; t.c = coef_div(a->c, b->c); t.m = mono_div(&a->m, &b->m); return t; } /* h:61 */
/* cpp:196-202 */
static poly poly_term_mul(const term* t, const poly* f) {
  poly g; poly_init(&g);
  g.sug = t->m.deg + f->sug;
  for (int k = 0; k < f->n; k++) poly_push(&g, term_mul(t, &f->t[k]));
  return g;
}
/* cpp:205-210 */
static poly poly_mul(const poly* f1, const poly* f2) {
  poly g; poly_init(&g);
  for (int k = 0; k < f1->n; k++) {
    poly tf = poly_term_mul(&f1->t[k], f2);
    poly s = poly_add(&g, &tf);
    poly_free(&tf); poly_free(&g);
    g = s;
  }
  return g;
}
typedef struct { poly* p; int n, cap; } polyvec;
static void pv_init(polyvec* v) { v->p = NULL; v->n = 0; v->cap = 0; }
static void pv_push(polyvec* v, poly f) { /* takes ownership */
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 8; v->p = (poly*)realloc(v->p, (size_t)v->cap * sizeof(poly)); }
  v->p[v->n++] = f;
}
static void pv_clear(polyvec* v) { for (int i = 0; i < v->n; i++) poly_free(&v->p[i]); v->n = 0; }
static void pv_free(polyvec* v) { pv_clear(v); free(v->p); pv_init(v); }
static polyvec pv_clone(const polyvec* v) {
  polyvec w; pv_init(&w);
  for (int i = 0; i < v->n; i++) pv_push(&w, poly_clone(&v->p[i]));
  return w;
}

/* ---- the string parser, polynomials.cpp:226-300 (needed by make_strat only) */
typedef struct { const char* s; } pstream;
static int ps_peek(pstream* p) { return *p->s ? (unsigned char)*p->s : -1; }
static int ps_get(pstream* p) { return *p->s ? (unsigned char)*p->s++ : -1; }
static int ps_int(pstream* p) { int v = 0; while (*p->s >= '0' && *p->s <= '9') v = v * 10 + (*p->s++ - '0'); return v; }
static mono parse_monomial(pstream* p) { /* cpp:233-257 */
  if (ps_peek(p) < 0) return mono_one();
  int var = ps_get(p) - 'a';
  int e[BO_N] = {0};
  if (var < 0 || var >= BO_N) return mono_one();
  int c = ps_peek(p);
  if (c == '^') {
    ps_get(p);
    e[var] += ps_int(p);
    mono m = mono_make(e);
    if (ps_peek(p) == '*') { ps_get(p); mono r = parse_monomial(p); return mono_mul(&m, &r); }
    return m;
  } else if (c == '*') {
    ps_get(p);
    e[var] += 1;
    mono m = mono_make(e); mono r = parse_monomial(p);
    return mono_mul(&m, &r);
  }
  e[var] += 1;
  return mono_make(e);
}
static term parse_term(pstream* p) { /* cpp:260-284 */
  int c = ps_peek(p);
  if (c == '+') { ps_get(p); return parse_term(p); }
  if (c == '-') { ps_get(p); term m1; m1.c = coef_norm(-1); m1.m = mono_one(); term t = parse_term(p); return term_mul(&m1, &t); }
  term t;
  if (c >= '0' && c <= '9') {
    t.c = coef_norm(ps_int(p));
    if (ps_peek(p) == '*') { ps_get(p); t.m = parse_monomial(p); } else t.m = mono_one();
  } else { t.c = 1; t.m = parse_monomial(p); }
  return t;
}
static poly parse_poly(pstream* p) { /* cpp:287-294 */
  poly f; poly_init(&f);
  while (ps_peek(p) >= 0) { /* Polynomial{t} + parse_polynomial(rest): right fold of commutative merges */
    term t = parse_term(p);
    poly s = poly_single(t);
    poly g = poly_add(&f, &s);
    poly_free(&f); poly_free(&s);
    f = g;
  }
  return f;
}

/* ======================================================================== */
/* spoly / reduce / update                           buchberger.cpp:18-99    */
/* ======================================================================== */

/* buchberger.cpp:18-21 */
static poly spoly(const poly* f, const poly* g) {
  term gamma; gamma.c = 1; gamma.m = mono_lcm(&f->t[0].m, &g->t[0].m);
  term tf = term_div(&gamma, &f->t[0]);
  term tg = term_div(&gamma, &g->t[0]);
  poly a = poly_term_mul(&tf, f), b = poly_term_mul(&tg, g);
  poly s = poly_sub(&a, &b);
  poly_free(&a); poly_free(&b);
  return s;
}

/* per-call traffic accounting for the SURVEY 8(d) algorithmic-bytes formula */
typedef struct { long long lm_scanned, f_terms, h_terms; } reduce_acct;

/* buchberger.cpp:24-49: full (head+tail) division; first divisor in the order
 * of F wins; steps counts successful reductions only */
/* test statistic: the longest polynomial a reduction of this thread has held (tests pick inputs that outgrow the
 * device's starting capacities with it); bo_stat_max_terms(1) reads and clears it */
static __thread int bo_max_terms = 0;
int bo_stat_max_terms(int reset) { const int v = bo_max_terms; if (reset) bo_max_terms = 0; return v; }

static poly reduce(const poly* g, const poly* const* F, int nF, int* steps_out, reduce_acct* acct) {
  int steps = 0;
  poly r; poly_init(&r);
  poly h = poly_clone(g);
  while (h.n != 0) {
    if (h.n > bo_max_terms) bo_max_terms = h.n;
    int found = 0;
    for (int k = 0; k < nF; k++) {
      const poly* f = F[k];
      if (mono_divisible(&h.t[0].m, &f->t[0].m)) {
        term q = term_div(&h.t[0], &f->t[0]);
        poly qf = poly_term_mul(&q, f);
        poly nh = poly_sub(&h, &qf);
        if (acct) { acct->lm_scanned += k + 1; acct->f_terms += f->n; acct->h_terms += h.n + nh.n; }
        poly_free(&qf); poly_free(&h);
        h = nh;
        found = 1; steps++;
        break;
      }
    }
    if (!found) {
      poly lt = poly_single(h.t[0]);
      poly nr = poly_add(&r, &lt);
      poly nh = poly_sub(&h, &lt);
      if (acct) { acct->lm_scanned += nF; acct->h_terms += h.n + nh.n; }
      poly_free(&r); poly_free(&h); poly_free(&lt);
      r = nr; h = nh;
    }
  }
  poly res = poly_add(&r, &h);
  poly_free(&r); poly_free(&h);
  *steps_out = steps;
  return res;
}

typedef struct { int i, j; } spair;
typedef struct { spair* p; int n, cap; } pairvec;
static void pr_init(pairvec* v) { v->p = NULL; v->n = 0; v->cap = 0; }
static void pr_push(pairvec* v, spair s) {
  if (v->n == v->cap) { v->cap = v->cap ? 2 * v->cap : 16; v->p = (spair*)realloc(v->p, (size_t)v->cap * sizeof(spair)); }
  v->p[v->n++] = s;
}
static void pr_free(pairvec* v) { free(v->p); pr_init(v); }
static pairvec pr_clone(const pairvec* v) { pairvec w; pr_init(&w); for (int k = 0; k < v->n; k++) pr_push(&w, v->p[k]); return w; }

enum { ELIM_GM = 0, ELIM_LCM = 1, ELIM_NONE = 2 }; /* buchberger.h:58 order */
enum { REW_ADDITIONS = 0, REW_REDUCTIONS = 1 };    /* buchberger.h:93 */

/* buchberger.cpp:52-99.  f is copied into G. */
static void update(polyvec* G, pairvec* P, const poly* f, int elim) {
  int m = G->n;
  const mono* lmf = &f->t[0].m;
  pairvec P_; pr_init(&P_);
  if (elim == ELIM_NONE) { /* 58-62 */
    for (int i = 0; i < m; i++) { spair s = {i, m}; pr_push(&P_, s); }
  } else if (elim == ELIM_LCM) { /* 63-68 */
    for (int i = 0; i < m; i++) {
      mono l = mono_lcm(&G->p[i].t[0].m, lmf), pr = mono_mul(&G->p[i].t[0].m, lmf);
      if (!mono_eq(&l, &pr)) { spair s = {i, m}; pr_push(&P_, s); }
    }
  } else { /* 69-95 Gebauer-Moeller */
    /* 70-76: stable removal of old pairs */
    int w = 0;
    for (int k = 0; k < P->n; k++) {
      spair p = P->p[k];
      mono l = mono_lcm(&G->p[p.i].t[0].m, &G->p[p.j].t[0].m);
      mono li = mono_lcm(&G->p[p.i].t[0].m, lmf), lj = mono_lcm(&G->p[p.j].t[0].m, lmf);
      int drop = mono_divisible(&l, lmf) && !mono_eq(&l, &li) && !mono_eq(&l, &lj);
      if (!drop) P->p[w++] = p;
    }
    P->n = w;
    /* 78-81: std::map<Monomial, vector<int>> == distinct lcms in ascending
     * order, each with its indices in increasing i */
    mono* keys = (mono*)malloc((size_t)(m ? m : 1) * sizeof(mono));
    int* first = (int*)malloc((size_t)(m ? m : 1) * sizeof(int));   /* v[0] of each bucket */
    char* coprime = (char*)calloc((size_t)(m ? m : 1), 1);          /* any i in bucket coprime to f */
    int nk = 0;
    for (int i = 0; i < m; i++) {
      const mono* lmi = &G->p[i].t[0].m;
      mono l = mono_lcm(lmi, lmf), pr = mono_mul(lmi, lmf);
      int cp = mono_eq(&l, &pr);
      int pos = 0;
      while (pos < nk && mono_lt(&keys[pos], &l)) pos++;
      if (pos < nk && !mono_lt(&l, &keys[pos])) { /* equivalent key */
        if (cp) coprime[pos] = 1;
      } else {
        memmove(keys + pos + 1, keys + pos, (size_t)(nk - pos) * sizeof(mono));
        memmove(first + pos + 1, first + pos, (size_t)(nk - pos) * sizeof(int));
        memmove(coprime + pos + 1, coprime + pos, (size_t)(nk - pos));
        keys[pos] = l; first[pos] = i; coprime[pos] = (char)cp;
        nk++;
      }
    }
    /* 82-91 */
    mono* min_lcms = (mono*)malloc((size_t)(nk ? nk : 1) * sizeof(mono));
    int nmin = 0;
    for (int b = 0; b < nk; b++) {
      int ok = 1;
      for (int q = 0; q < nmin; q++) if (mono_divisible(&keys[b], &min_lcms[q])) { ok = 0; break; }
      if (ok) {
        min_lcms[nmin++] = keys[b];
        if (!coprime[b]) { spair s = {first[b], m}; pr_push(&P_, s); }
      }
    }
    /* 92: sort by i (keys distinct -> any correct sort gives the same result) */
    for (int a = 1; a < P_.n; a++) {
      spair s = P_.p[a]; int b = a - 1;
      while (b >= 0 && P_.p[b].i > s.i) { P_.p[b + 1] = P_.p[b]; b--; }
      P_.p[b + 1] = s;
    }
    free(keys); free(first); free(coprime); free(min_lcms);
  }
  pv_push(G, poly_clone(f));                         /* 97 */
  for (int k = 0; k < P_.n; k++) pr_push(P, P_.p[k]); /* 98 */
  pr_free(&P_);
}

/* ======================================================================== */
/* minimalize / interreduce / buchberger            buchberger.cpp:102-266   */
/* ======================================================================== */
static int lm_asc_cmp(void* ctx, int a, int b) { /* f.LM() < g.LM() */
  polyvec* v = (polyvec*)ctx;
  return mono_lt(&v->p[a].t[0].m, &v->p[b].t[0].m);
}
/* buchberger.cpp:102-112 */
static polyvec minimalize(const polyvec* G) {
  polyvec out; pv_init(&out);
  int* idx = (int*)malloc((size_t)(G->n ? G->n : 1) * sizeof(int));
  for (int i = 0; i < G->n; i++) idx[i] = i;
  std_sort(idx, G->n, lm_asc_cmp, (void*)G);
  for (int a = 0; a < G->n; a++) {
    const poly* g = &G->p[idx[a]];
    int ok = 1;
    for (int b = 0; b < out.n; b++) if (mono_divisible(&g->t[0].m, &out.p[b].t[0].m)) { ok = 0; break; }
    if (ok) pv_push(&out, poly_clone(g));
  }
  free(idx);
  return out;
}
/* buchberger.cpp:115-122 */
static polyvec interreduce(const polyvec* G) {
  polyvec out; pv_init(&out);
  const poly** F = (const poly**)malloc((size_t)(G->n ? G->n : 1) * sizeof(poly*));
  for (int i = 0; i < G->n; i++) F[i] = &G->p[i];
  for (int i = 0; i < G->n; i++) {
    const poly* g = &G->p[i];
    term t; t.c = coef_div(1, g->t[0].c); t.m = mono_one();
    poly lt = poly_single(g->t[0]);
    poly tail = poly_sub(g, &lt);
    int steps;
    poly r = reduce(&tail, F, G->n, &steps, NULL);
    poly s = poly_add(&r, &lt);
    pv_push(&out, poly_term_mul(&t, &s));
    poly_free(&lt); poly_free(&tail); poly_free(&r); poly_free(&s);
  }
  free(F);
  return out;
}

/* ---- libstdc++ <random> restated (GCC 11 bits/random.h, random.tcc, uniform_int_dist.h) */
typedef struct { uint64_t x; } minstd0; /* std::default_random_engine = minstd_rand0 */
static void rng_seed(minstd0* r, int seed) {
  /* linear_congruential_engine::seed(result_type): c == 0, so x = s mod m, 0 -> 1 */
  uint64_t s = (uint64_t)(long long)seed; /* int -> unsigned long conversion */
  r->x = s % 2147483647ull;
  if (r->x == 0) r->x = 1;
}
static uint64_t rng_next(minstd0* r) { r->x = (r->x * 16807ull) % 2147483647ull; return r->x; }
#define RNG_MIN 1ull
#define RNG_MAX 2147483646ull
/* uniform_int_distribution<int>(a,b)(urng): the "downscaling, two divisions" branch */
static int rng_uniform_int(minstd0* r, int a, int b) {
  const uint64_t urngrange = RNG_MAX - RNG_MIN;
  const uint64_t urange = (uint64_t)((long long)b - (long long)a);
  uint64_t ret;
  if (urngrange > urange) {
    const uint64_t uerange = urange + 1;
    const uint64_t scaling = urngrange / uerange;
    const uint64_t past = uerange * scaling;
    do ret = rng_next(r) - RNG_MIN; while (ret >= past);
    ret /= scaling;
  } else {
    ret = rng_next(r) - RNG_MIN; /* (upscaling branch not reachable for int ranges here) */
  }
  return (int)(ret + (uint64_t)(long long)a);
}
/* generate_canonical<double, 53>(urng): k = 2 draws for minstd_rand0 */
static double rng_canonical(minstd0* r) {
  const long double R = (long double)RNG_MAX - (long double)RNG_MIN + 1.0L;
  double sum = 0.0, tmp = 1.0;
  for (int k = 2; k != 0; --k) {
    sum += (double)(rng_next(r) - RNG_MIN) * tmp;
    tmp = (double)((long double)tmp * R);
  }
  double ret = sum / tmp;
  if (ret >= 1.0) ret = nextafter(1.0, 0.0);
  return ret;
}
typedef struct { double* prob; double* cp; int n; } discrete_dist;
/* discrete_distribution::param_type::_M_initialize */
static discrete_dist dd_make(const int* w, int n) {
  discrete_dist d; d.prob = NULL; d.cp = NULL; d.n = 0;
  if (n < 2) return d;
  d.n = n;
  d.prob = (double*)malloc((size_t)n * sizeof(double));
  d.cp = (double*)malloc((size_t)n * sizeof(double));
  double sum = 0.0;
  for (int i = 0; i < n; i++) sum += (double)w[i];
  for (int i = 0; i < n; i++) d.prob[i] = (double)w[i] / sum;
  double acc = 0.0;
  for (int i = 0; i < n; i++) { acc += d.prob[i]; d.cp[i] = acc; } /* std::partial_sum */
  d.cp[n - 1] = 1.0;
  return d;
}
static discrete_dist dd_clone(const discrete_dist* s) {
  discrete_dist d = *s;
  if (s->n) {
    d.prob = (double*)malloc((size_t)s->n * sizeof(double)); memcpy(d.prob, s->prob, (size_t)s->n * sizeof(double));
    d.cp = (double*)malloc((size_t)s->n * sizeof(double)); memcpy(d.cp, s->cp, (size_t)s->n * sizeof(double));
  }
  return d;
}
static void dd_free(discrete_dist* d) { free(d->prob); free(d->cp); d->prob = d->cp = NULL; d->n = 0; }
static int dd_draw(const discrete_dist* d, minstd0* r) {
  if (d->n == 0) return 0;
  double p = rng_canonical(r);
  int lo = 0, hi = d->n; /* std::lower_bound(cp, p) */
  while (lo < hi) { int mid = lo + (hi - lo) / 2; if (d->cp[mid] < p) lo = mid + 1; else hi = mid; }
  return lo;
}
/* std::poisson_distribution<int>(mean) of libstdc++ 11 (bits/random.tcc:1261-1404) over minstd_rand0, both branches:
 * products of canonical draws below mean 12, Devroye's rejection algorithm from 12 on, with the distribution's own
 * normal_distribution<double> (polar method; the second variate of a pair is kept for the next call). */
typedef struct {
  double mean, lm_thr, lfm, sm, d, scx, cx1, c2b, cb;
  int saved_available; double saved;
} poisson_dist;
static void pd_init(poisson_dist* p, double mean) {
  memset(p, 0, sizeof *p);
  p->mean = mean;
  if (mean >= 12) {
    const double m = floor(mean);
    p->lm_thr = log(mean); p->lfm = lgamma(m + 1); p->sm = sqrt(m);
    const double pi_4 = 0.7853981633974483096156608458198757L;
    const double dx = sqrt(2 * m * log(32 * m / pi_4));
    double dd = m < dx ? m : dx; if (dd < 6.0) dd = 6.0;
    p->d = round(dd);
    const double cx = 2 * m + p->d;
    p->scx = sqrt(cx / 2); p->cx1 = 1 / cx;
    p->c2b = sqrt(pi_4 * cx) * exp(p->cx1);
    p->cb = 2 * cx * exp(-p->d * p->cx1 * (1 + p->d / 2)) / p->d;
  } else p->lm_thr = exp(-mean);
}
static double pd_normal(poisson_dist* p, minstd0* r) {
  double ret;
  if (p->saved_available) { p->saved_available = 0; ret = p->saved; }
  else {
    double x, y, r2;
    do { x = 2.0 * rng_canonical(r) - 1.0; y = 2.0 * rng_canonical(r) - 1.0; r2 = x * x + y * y; } while (r2 > 1.0 || r2 == 0.0);
    const double mult = sqrt(-2 * log(r2) / r2);
    p->saved = x * mult; p->saved_available = 1;
    ret = y * mult;
  }
  return ret * 1.0 + 0.0;
}
static int rng_poisson(minstd0* r, poisson_dist* p) {
  if (p->mean < 12) {
    int x = 0;
    double prod = 1.0;
    do { prod *= rng_canonical(r); x += 1; } while (prod > p->lm_thr);
    return x - 1;
  }
  double x;
  const double naf = (1 - 2.220446049250313e-16) / 2;
  const double thr = 2147483647.0 + naf;
  const double m = floor(p->mean);
  const double spi_2 = 1.2533141373155002512078826424055226L;
  const double c1 = p->sm * spi_2, c2 = p->c2b + c1, c3 = c2 + 1, c4 = c3 + 1;
  const double k178 = 0.0128205128205128205128205128205128L, e178 = 1.0129030479320018583185514777512983L;
  const double c5 = c4 + e178, c = p->cb + c5, cx2 = 2 * (2 * m + p->d);
  int reject = 1;
  do {
    const double u = c * rng_canonical(r);
    const double e = -log(1.0 - rng_canonical(r));
    double w = 0.0;
    if (u <= c1) {
      const double n = pd_normal(p, r);
      const double y = -fabs(n) * p->sm - 1;
      x = floor(y);
      w = -n * n / 2;
      if (x < -m) continue;
    } else if (u <= c2) {
      const double n = pd_normal(p, r);
      const double y = 1 + fabs(n) * p->scx;
      x = ceil(y);
      w = y * (2 - y) * p->cx1;
      if (x > p->d) continue;
    } else if (u <= c3) x = -1;
    else if (u <= c4) x = 0;
    else if (u <= c5) { x = 1; w = k178; }
    else {
      const double v = -log(1.0 - rng_canonical(r));
      const double y = p->d + v * cx2 / p->d;
      x = ceil(y);
      w = -p->d * p->cx1 * (1 + y / 2);
    }
    reject = (w - e - x * p->lm_thr > p->lfm - lgamma(x + m + 1));
    reject |= x + m >= thr;
  } while (reject);
  return (int)(x + m + naf);
}

/* ---- selection strategies, buchberger.h:111 ------------------------------ */
enum { SEL_FIRST = 0, SEL_DEGREE, SEL_NORMAL, SEL_SUGAR, SEL_RANDOM, SEL_LAST, SEL_CODEGREE, SEL_STRANGE, SEL_SPICE };

static int pair_sugar(const polyvec* G, spair p, const mono* l) { /* cpp:191-192 */
  mono a = mono_div(l, &G->p[p.i].t[0].m), b = mono_div(l, &G->p[p.j].t[0].m);
  int s1 = G->p[p.i].sug + a.deg, s2 = G->p[p.j].sug + b.deg;
  return s1 > s2 ? s1 : s2;
}
/* lexicographic compare of (sugar?, lcm or degree, j, i); returns <0, 0, >0 */
static int pair_key_cmp(const polyvec* G, int sel, spair p1, spair p2) {
  int base = sel;
  if (sel == SEL_LAST) base = SEL_FIRST;
  if (sel == SEL_CODEGREE) base = SEL_DEGREE;
  if (sel == SEL_STRANGE) base = SEL_NORMAL;
  if (sel == SEL_SPICE) base = SEL_SUGAR;
  if (base != SEL_FIRST) {
    mono m1 = mono_lcm(&G->p[p1.i].t[0].m, &G->p[p1.j].t[0].m);
    mono m2 = mono_lcm(&G->p[p2.i].t[0].m, &G->p[p2.j].t[0].m);
    if (base == SEL_SUGAR) {
      int s1 = pair_sugar(G, p1, &m1), s2 = pair_sugar(G, p2, &m2);
      if (s1 != s2) return s1 < s2 ? -1 : 1;
    }
    if (base == SEL_DEGREE) {
      if (m1.deg != m2.deg) return m1.deg < m2.deg ? -1 : 1;
    } else { /* NORMAL / SUGAR compare the monomials with operator< of std::tie */
      if (mono_lt(&m1, &m2)) return -1;
      if (mono_lt(&m2, &m1)) return 1;
    }
  }
  if (p1.j != p2.j) return p1.j < p2.j ? -1 : 1;
  if (p1.i != p2.i) return p1.i < p2.i ? -1 : 1;
  return 0;
}
static int pair_select_less(const polyvec* G, int sel, spair p1, spair p2) { /* cpp:165-240 */
  int c = pair_key_cmp(G, sel, p1, p2);
  if (sel == SEL_LAST || sel == SEL_CODEGREE || sel == SEL_STRANGE || sel == SEL_SPICE) return c > 0;
  return c < 0;
}

typedef struct { int zero_reductions, nonzero_reductions, polynomial_additions; double total_reward, discounted_return; } bstats;

static void gord_insert_sorted(int** ord, int* n, int* cap, const polyvec* G, int gi) {
  /* std::upper_bound on LM, then insert: buchberger.cpp:257, 309, 324 */
  const mono* lm = &G->p[gi].t[0].m;
  int lo = 0, hi = *n;
  while (lo < hi) { int mid = lo + (hi - lo) / 2; if (mono_lt(lm, &G->p[(*ord)[mid]].t[0].m)) hi = mid; else lo = mid + 1; }
  if (*n == *cap) { *cap = *cap ? 2 * *cap : 16; *ord = (int*)realloc(*ord, (size_t)*cap * sizeof(int)); }
  memmove(*ord + lo + 1, *ord + lo, (size_t)(*n - lo) * sizeof(int));
  (*ord)[lo] = gi;
  (*n)++;
}
static void gord_push(int** ord, int* n, int* cap, int gi) {
  if (*n == *cap) { *cap = *cap ? 2 * *cap : 16; *ord = (int*)realloc(*ord, (size_t)*cap * sizeof(int)); }
  (*ord)[(*n)++] = gi;
}

/* buchberger.cpp:143-266 (the (F, S) overload).  Takes ownership of nothing. */
static polyvec buchberger_pairs(const polyvec* F, const pairvec* S, int selection, int elim, int rewards,
                                int sort_reducers, double gamma, int has_seed, int seed, int want_basis, bstats* st) {
  polyvec G = pv_clone(F);
  pairvec P = pr_clone(S);
  bstats stats; memset(&stats, 0, sizeof stats);
  double discount = 1.0;
  /* reducers G_ as an index order into G */
  int* ord = NULL; int nord = 0, cord = 0;
  for (int i = 0; i < G.n; i++) gord_push(&ord, &nord, &cord, i);
  if (sort_reducers) std_sort(ord, nord, lm_asc_cmp, &G); /* 157-158: std::sort, NOT stable */
  minstd0 rng; rng.x = 1;
  if (selection == SEL_RANDOM) rng_seed(&rng, has_seed ? seed : (int)time(NULL)); /* 200-204 */
  const poly** Fp = NULL; int cF = 0;
  while (P.n != 0) {
    int pick = 0;
    if (selection == SEL_RANDOM) pick = rng_uniform_int(&rng, 0, P.n - 1); /* choice(): ideals.h:68-73 */
    else for (int k = 1; k < P.n; k++) if (pair_select_less(&G, selection, P.p[k], P.p[pick])) pick = k; /* min_element */
    spair p = P.p[pick];
    memmove(P.p + pick, P.p + pick + 1, (size_t)(P.n - pick - 1) * sizeof(spair)); P.n--;
    if (cF < nord) { cF = nord + 16; Fp = (const poly**)realloc((void*)Fp, (size_t)cF * sizeof(poly*)); }
    for (int k = 0; k < nord; k++) Fp[k] = &G.p[ord[k]];
    poly s = spoly(&G.p[p.i], &G.p[p.j]);
    int steps;
    poly r = reduce(&s, Fp, nord, &steps, NULL);
    poly_free(&s);
    double reward = (rewards == REW_ADDITIONS) ? (-1.0 - steps) : -1.0; /* 248 */
    stats.polynomial_additions += steps + 1;
    stats.total_reward += reward;
    stats.discounted_return += discount * reward;
    discount *= gamma;
    if (r.n != 0) {
      update(&G, &P, &r, elim);
      stats.nonzero_reductions++;
      if (sort_reducers) gord_insert_sorted(&ord, &nord, &cord, &G, G.n - 1);
      else gord_push(&ord, &nord, &cord, G.n - 1);
    } else stats.zero_reductions++;
    poly_free(&r);
  }
  free((void*)Fp); free(ord); pr_free(&P);
  *st = stats;
  if (!want_basis) { pv_free(&G); polyvec e; pv_init(&e); return e; }
  polyvec mn = minimalize(&G);
  polyvec red = interreduce(&mn); /* 265 */
  pv_free(&G); pv_free(&mn);
  return red;
}
/* buchberger.cpp:125-140 (the F-only overload; NB sort_input is accepted and ignored there) */
static polyvec buchberger_gens(const polyvec* F, int selection, int elim, int rewards, int sort_reducers,
                               double gamma, int has_seed, int seed, int want_basis, bstats* st) {
  polyvec G; pv_init(&G);
  pairvec P; pr_init(&P);
  for (int i = 0; i < F->n; i++) update(&G, &P, &F->p[i], elim);
  polyvec res = buchberger_pairs(&G, &P, selection, elim, rewards, sort_reducers, gamma, has_seed, seed, want_basis, st);
  pv_free(&G); pr_free(&P);
  return res;
}

/* ======================================================================== */
/* Ideal generators                                  ideals.h / ideals.cpp   */
/* ======================================================================== */
/* ideals.cpp:16-36 */
static polyvec cyclic_ideal(int n) {
  polyvec F; pv_init(&F);
  for (int d = 1; d < n; d++) {
    term* ts = (term*)malloc((size_t)n * sizeof(term));
    for (int i = 0; i < n; i++) {
      int e[BO_N] = {0};
      for (int k = 0; k < d; k++) e[(i + k) % n] = 1;
      ts[i].c = 1; ts[i].m = mono_make(e);
    }
    pv_push(&F, poly_from_terms(ts, n));
    free(ts);
  }
  int e[BO_N] = {0};
  for (int i = 0; i < n; i++) e[i] = 1;
  term two[2];
  two[0].c = 1; two[0].m = mono_make(e);
  two[1].c = coef_norm(-1); two[1].m = mono_one();
  pv_push(&F, poly_from_terms(two, 2));
  return F;
}
/* std::next_permutation on an int array */
static int next_permutation(int* a, int n) {
  if (n < 2) return 0;
  int i = n - 1;
  for (;;) {
    int ii = i; --i;
    if (a[i] < a[ii]) {
      int j = n - 1;
      while (!(a[i] < a[j])) --j;
      int t = a[i]; a[i] = a[j]; a[j] = t;
      for (int l = ii, r = n - 1; l < r; l++, r--) { t = a[l]; a[l] = a[r]; a[r] = t; }
      return 1;
    }
    if (i == 0) {
      for (int l = 0, r = n - 1; l < r; l++, r--) { int t = a[l]; a[l] = a[r]; a[r] = t; }
      return 0;
    }
  }
}
typedef struct { mono* m; int n; } monovec;
/* ideals.cpp:39-64: stars (0) and bars (1), all permutations in next_permutation order */
static monovec basis(int n, int d) {
  int len = d + n - 1;
  int* a = (int*)malloc((size_t)(len ? len : 1) * sizeof(int));
  for (int i = 0; i < d; i++) a[i] = 0;
  for (int i = 0; i < n - 1; i++) a[d + i] = 1;
  monovec B; B.m = NULL; B.n = 0; int cap = 0;
  do {
    int e[BO_N] = {0};
    int index = 0;
    for (int i = 0; i < len; i++) { if (a[i] == 0) e[index]++; else index++; }
    if (B.n == cap) { cap = cap ? 2 * cap : 64; B.m = (mono*)realloc(B.m, (size_t)cap * sizeof(mono)); }
    B.m[B.n++] = mono_make(e);
  } while (next_permutation(a, len));
  free(a);
  return B;
}
static int binomial(int n, int k) { /* ideals.cpp:67-72 */
  if (k == 0 || k == n) return 1;
  return binomial(n - 1, k - 1) + binomial(n - 1, k);
}
enum { DIST_UNIFORM = 0, DIST_WEIGHTED = 1, DIST_MAXIMUM = 2 }; /* ideals.h:48 */
/* ideals.cpp:75-100: weights */
static int degree_weights(int n, int d, int dist, int constants, int* w) {
  int k = 0;
  w[k++] = constants ? 1 : 0;
  if (dist == DIST_UNIFORM) for (int i = 1; i < d + 1; i++) w[k++] = binomial(n + i - 1, n - 1);
  else if (dist == DIST_WEIGHTED) for (int i = 0; i < d; i++) w[k++] = 1;
  else { for (int i = 0; i < d - 1; i++) w[k++] = 0; w[k++] = 1; }
  return k;
}

enum { GEN_FIXED = 0, GEN_BINOMIAL = 1, GEN_RANDOM = 2 };
typedef struct {
  int kind, n, s, d;
  int homogeneous, pure;
  double lam, lm_thr;
  poisson_dist length_dist;
  monovec* bases; /* bases[0..d] */
  discrete_dist degree_dist;
  minstd0 rng;
  polyvec F; /* fixed */
} gen;

static void gen_free(gen* g) {
  if (!g) return;
  if (g->bases) { for (int i = 0; i <= g->d; i++) free(g->bases[i].m); free(g->bases); }
  dd_free(&g->degree_dist);
  pv_free(&g->F);
  free(g);
}
static gen* gen_clone(const gen* s) {
  gen* g = (gen*)malloc(sizeof(gen));
  *g = *s;
  if (s->bases) {
    g->bases = (monovec*)malloc((size_t)(s->d + 1) * sizeof(monovec));
    for (int i = 0; i <= s->d; i++) {
      g->bases[i].n = s->bases[i].n;
      g->bases[i].m = (mono*)malloc((size_t)s->bases[i].n * sizeof(mono));
      memcpy(g->bases[i].m, s->bases[i].m, (size_t)s->bases[i].n * sizeof(mono));
    }
  }
  g->degree_dist = dd_clone(&s->degree_dist);
  g->F = pv_clone(&s->F);
  return g;
}
static gen* gen_fixed(const polyvec* F) { /* ideals.cpp:146-154: n = max variable INDEX (sic) */
  gen* g = (gen*)calloc(1, sizeof(gen));
  g->kind = GEN_FIXED;
  g->F = pv_clone(F);
  g->n = 0;
  for (int i = 0; i < F->n; i++)
    for (int k = 0; k < F->p[i].n; k++)
      for (int x = 0; x < BO_N; x++)
        if (F->p[i].t[k].m.e[x] != 0 && x > g->n) g->n = x;
  return g;
}
static gen* gen_random_common(int kind, int n, int d, int s, int dist, int constants, int homogeneous) {
  gen* g = (gen*)calloc(1, sizeof(gen));
  g->kind = kind; g->n = n; g->d = d; g->s = s; g->homogeneous = homogeneous;
  g->bases = (monovec*)malloc((size_t)(d + 1) * sizeof(monovec));
  for (int i = 0; i < d + 1; i++) g->bases[i] = basis(n, i); /* ideals.cpp:160-161, 206-207 */
  int* w = (int*)malloc((size_t)(d + 2) * sizeof(int));
  int nw = degree_weights(n, d, dist, constants, w);
  g->degree_dist = dd_make(w, nw);
  free(w);
  rng_seed(&g->rng, (int)time(NULL)); /* reference seeds from random_device; callers seed explicitly */
  return g;
}
static int split_dash(const char* s, char out[][32], int max) {
  int n = 0;
  while (*s && n < max) {
    int k = 0;
    while (*s && *s != '-') { if (k < 31) out[n][k++] = *s; s++; }
    out[n][k] = 0; n++;
    if (*s == '-') s++;
  }
  return n;
}
static int dist_type_of(const char* s) {
  if (!strcmp(s, "uniform")) return DIST_UNIFORM;
  if (!strcmp(s, "weighted")) return DIST_WEIGHTED;
  if (!strcmp(s, "maximum")) return DIST_MAXIMUM;
  return -1;
}
/* ideals.cpp:103-143 */
static gen* parse_ideal_dist(const char* dist) {
  char a[12][32];
  int na = split_dash(dist, a, 12);
  int has = 0;
  if (na >= 2 && !strcmp(a[0], "cyclic")) {
    polyvec F = cyclic_ideal(atoi(a[1]));
    gen* g = gen_fixed(&F);
    pv_free(&F);
    return g;
  }
  if (na < 4) return NULL;
  int consts = 0, homog = 0, pure = 0;
  for (int i = 0; i < na; i++) {
    if (!strcmp(a[i], "consts")) consts = 1;
    if (!strcmp(a[i], "homog")) homog = 1;
    if (!strcmp(a[i], "pure")) pure = 1;
  }
  (void)has;
  if (dist_type_of(a[3]) >= 0) {
    gen* g = gen_random_common(GEN_BINOMIAL, atoi(a[0]), atoi(a[1]), atoi(a[2]), dist_type_of(a[3]), consts, homog);
    g->pure = pure;
    return g;
  }
  if (na < 5) return NULL;
  int dt = dist_type_of(a[4]);
  if (dt < 0) dt = DIST_UNIFORM; /* dist_types[unknown] default-inserts Uniform (=0) */
  gen* g = gen_random_common(GEN_RANDOM, atoi(a[0]), atoi(a[1]), atoi(a[2]), dt, consts, homog);
  g->lam = atof(a[3]);
  g->lm_thr = exp(-g->lam);
  pd_init(&g->length_dist, g->lam);
  return g;
}
static mono gen_choice(gen* g, int d) { /* choice(): fresh uniform_int_distribution(0, len-1), ideals.h:68-73 */
  int k = rng_uniform_int(&g->rng, 0, g->bases[d].n - 1);
  return g->bases[d].m[k];
}
/* ideals.cpp:168-201 and 214-231.  Returns 0 on the reference's runtime_error. */
static int gen_next(gen* g, polyvec* F) {
  pv_clear(F);
  if (g->kind == GEN_FIXED) {
    for (int i = 0; i < g->F.n; i++) pv_push(F, poly_clone(&g->F.p[i]));
    return 1;
  }
  if (g->kind == GEN_BINOMIAL) {
    for (int i = 0; i < g->s; i++) {
      int c = g->pure ? coef_norm(-1) : rng_uniform_int(&g->rng, 1, BO_P - 1);
      int d1, d2;
      if (g->homogeneous) d1 = d2 = dd_draw(&g->degree_dist, &g->rng);
      else { d1 = dd_draw(&g->degree_dist, &g->rng); d2 = dd_draw(&g->degree_dist, &g->rng); }
      int success = 0;
      for (int trials = 0; trials < 1000; trials++) {
        mono m1 = gen_choice(g, d1), m2 = gen_choice(g, d2);
        term ts[2];
        if (mono_lt(&m1, &m2)) { ts[0].c = 1; ts[0].m = m2; ts[1].c = c; ts[1].m = m1; }
        else if (mono_gt(&m1, &m2)) { ts[0].c = 1; ts[0].m = m1; ts[1].c = c; ts[1].m = m2; }
        else continue;
        pv_push(F, poly_from_terms(ts, 2));
        success = 1;
        break;
      }
      if (!success) return 0;
    }
    return 1;
  }
  for (int i = 0; i < g->s; i++) { /* GEN_RANDOM */
    poly f; poly_init(&f);
    int terms = 2 + rng_poisson(&g->rng, &g->length_dist);
    int d = dd_draw(&g->degree_dist, &g->rng);
    for (int j = 0; j < terms; j++) {
      term t; t.c = rng_uniform_int(&g->rng, 1, BO_P - 1);
      t.m = gen_choice(g, d);
      poly s = poly_single(t);
      poly nf = poly_add(&f, &s);
      poly_free(&f); poly_free(&s);
      f = nf;
      if (!g->homogeneous) d = dd_draw(&g->degree_dist, &g->rng);
    }
    if (f.n == 0) { poly_free(&f); return 0; } /* reference: f.LC() on an empty polynomial is UB */
    term t; t.c = coef_div(1, f.t[0].c); t.m = mono_one();
    pv_push(F, poly_term_mul(&t, &f));
    poly_free(&f);
  }
  return 1;
}

/* ======================================================================== */
/* BuchbergerEnv                                    buchberger.cpp:269-351   */
/* ======================================================================== */
typedef struct {
  gen* g;
  int elim, rewards, sort_input, sort_reducers;
  polyvec G;
  pairvec P;
  int* ord; int nord, cord; /* G_ : reducers as indices into G */
  long long last_bytes;
} env;

static void env_reset(env* e) { /* 299-315 */
  for (;;) {
    polyvec F; pv_init(&F);
    gen_next(e->g, &F);
    int* idx = (int*)malloc((size_t)(F.n ? F.n : 1) * sizeof(int));
    for (int i = 0; i < F.n; i++) idx[i] = i;
    if (e->sort_input) std_sort(idx, F.n, lm_asc_cmp, &F); /* 301-302 */
    pv_clear(&e->G); e->P.n = 0; e->nord = 0;
    for (int a = 0; a < F.n; a++) {
      update(&e->G, &e->P, &F.p[idx[a]], e->elim);
      if (e->sort_reducers) gord_insert_sorted(&e->ord, &e->nord, &e->cord, &e->G, e->G.n - 1);
      else gord_push(&e->ord, &e->nord, &e->cord, e->G.n - 1);
    }
    free(idx); pv_free(&F);
    if (e->P.n != 0) break; /* 313-314: redraw */
  }
}
static double env_step(env* e, spair action) { /* 318-329 */
  int w = 0;
  for (int k = 0; k < e->P.n; k++) /* std::remove(action) */
    if (!(e->P.p[k].i == action.i && e->P.p[k].j == action.j)) e->P.p[w++] = e->P.p[k];
  e->P.n = w;
  const poly** F = (const poly**)malloc((size_t)(e->nord ? e->nord : 1) * sizeof(poly*));
  for (int k = 0; k < e->nord; k++) F[k] = &e->G.p[e->ord[k]];
  poly s = spoly(&e->G.p[action.i], &e->G.p[action.j]);
  reduce_acct acct = {0, 0, 0};
  int steps;
  poly r = reduce(&s, F, e->nord, &steps, &acct);
  long long bytes = 12LL * (e->G.p[action.i].n + e->G.p[action.j].n) + 12LL * s.n
                  + 8LL * acct.lm_scanned + 12LL * acct.f_terms + 12LL * acct.h_terms + 12LL * r.n;
  free((void*)F); poly_free(&s);
  if (r.n != 0) {
    long long pb = e->P.n, gb = e->G.n;
    update(&e->G, &e->P, &r, e->elim);
    bytes += 8LL * gb + 8LL * (pb + e->P.n);
    if (e->sort_reducers) gord_insert_sorted(&e->ord, &e->nord, &e->cord, &e->G, e->G.n - 1);
    else gord_push(&e->ord, &e->nord, &e->cord, e->G.n - 1);
  }
  poly_free(&r);
  e->last_bytes = bytes;
  return (e->rewards == REW_ADDITIONS) ? (-1.0 - steps) : -1.0;
}
static double env_value(const env* e, const char* strategy, double gamma) { /* 332-351 */
  bstats st;
  if (!strcmp(strategy, "sample")) {
    polyvec b = buchberger_pairs(&e->G, &e->P, SEL_DEGREE, e->elim, e->rewards, e->sort_reducers, gamma, 0, 0, 0, &st);
    pv_free(&b);
    double best = st.discounted_return;
    for (int i = 0; i < 100; i++) {
      b = buchberger_pairs(&e->G, &e->P, SEL_RANDOM, e->elim, e->rewards, e->sort_reducers, gamma, 0, 0, 0, &st);
      pv_free(&b);
      if (st.discounted_return > best) best = st.discounted_return;
    }
    return best;
  }
  int sel = SEL_FIRST; /* std::map::operator[] default-inserts SelectionType{0} == First for unknown keys */
  if (!strcmp(strategy, "first")) sel = SEL_FIRST;
  else if (!strcmp(strategy, "degree")) sel = SEL_DEGREE;
  else if (!strcmp(strategy, "normal")) sel = SEL_NORMAL;
  else if (!strcmp(strategy, "sugar")) sel = SEL_SUGAR;
  else if (!strcmp(strategy, "random")) sel = SEL_RANDOM;
  polyvec b = buchberger_pairs(&e->G, &e->P, sel, e->elim, e->rewards, e->sort_reducers, gamma, 0, 0, 0, &st);
  pv_free(&b);
  return st.discounted_return;
}

/* ======================================================================== */
/* exported C API                                                            */
/* ======================================================================== */
static term* terms_from_flat(int n, const int* coef, const int* exps) {
  term* ts = (term*)malloc((size_t)(n ? n : 1) * sizeof(term));
  for (int k = 0; k < n; k++) { ts[k].c = coef_norm(coef[k]); ts[k].m = mono_make(exps + k * BO_N); }
  return ts;
}
static void poly_to_flat(const poly* f, int* coef, int* exps) {
  for (int k = 0; k < f->n; k++) {
    coef[k] = f->t[k].c;
    for (int x = 0; x < BO_N; x++) exps[k * BO_N + x] = f->t[k].m.e[x];
  }
}
#define PL(x) ((polyvec*)(x))

void* bo_pl_new(void) { polyvec* v = (polyvec*)malloc(sizeof(polyvec)); pv_init(v); return v; }
void bo_pl_free(void* pl) { pv_free(PL(pl)); free(pl); }
void bo_pl_clear(void* pl) { pv_clear(PL(pl)); }
int bo_pl_len(void* pl) { return PL(pl)->n; }
void bo_pl_add(void* pl, int nterms, const int* coef, const int* exps) {
  term* ts = terms_from_flat(nterms, coef, exps);
  pv_push(PL(pl), poly_from_terms(ts, nterms));
  free(ts);
}
int bo_pl_nterms(void* pl, int i) { return PL(pl)->p[i].n; }
int bo_pl_sugar(void* pl, int i) { return PL(pl)->p[i].sug; }
void bo_pl_get(void* pl, int i, int* coef, int* exps) { poly_to_flat(&PL(pl)->p[i], coef, exps); }

int bo_coef_norm(int a) { return coef_norm(a); }
int bo_coef_add(int a, int b) { return coef_add(coef_norm(a), coef_norm(b)); }
int bo_coef_sub(int a, int b) { return coef_sub(coef_norm(a), coef_norm(b)); }
int bo_coef_mul(int a, int b) { return coef_mul(coef_norm(a), coef_norm(b)); }
int bo_coef_div(int a, int b) { return coef_div(coef_norm(a), coef_norm(b)); }
int bo_mono_gt(const int* a, const int* b) { mono x = mono_make(a), y = mono_make(b); return mono_gt(&x, &y); }

void bo_poly_add(void* pl, int i, int j, void* out) { pv_push(PL(out), poly_add(&PL(pl)->p[i], &PL(pl)->p[j])); }
void bo_poly_sub(void* pl, int i, int j, void* out) { pv_push(PL(out), poly_sub(&PL(pl)->p[i], &PL(pl)->p[j])); }
void bo_poly_mul(void* pl, int i, int j, void* out) { pv_push(PL(out), poly_mul(&PL(pl)->p[i], &PL(pl)->p[j])); }
void bo_parse_polynomial(const char* s, void* out) { pstream p; p.s = s; pv_push(PL(out), parse_poly(&p)); }
void bo_spoly(void* pl, int i, int j, void* out) { pv_push(PL(out), spoly(&PL(pl)->p[i], &PL(pl)->p[j])); }
int bo_reduce(void* plg, int gi, void* plF, void* out) {
  polyvec* F = PL(plF);
  const poly** Fp = (const poly**)malloc((size_t)(F->n ? F->n : 1) * sizeof(poly*));
  for (int k = 0; k < F->n; k++) Fp[k] = &F->p[k];
  int steps;
  pv_push(PL(out), reduce(&PL(plg)->p[gi], Fp, F->n, &steps, NULL));
  free((void*)Fp);
  return steps;
}
int bo_update(void* plG, int* pairs, int npairs, void* plf, int fi, int elim) {
  pairvec P; pr_init(&P);
  for (int k = 0; k < npairs; k++) { spair s = {pairs[2 * k], pairs[2 * k + 1]}; pr_push(&P, s); }
  update(PL(plG), &P, &PL(plf)->p[fi], elim);
  for (int k = 0; k < P.n; k++) { pairs[2 * k] = P.p[k].i; pairs[2 * k + 1] = P.p[k].j; }
  int n = P.n;
  pr_free(&P);
  return n;
}
void bo_minimalize(void* pl, void* out) { polyvec r = minimalize(PL(pl)); pv_free(PL(out)); *PL(out) = r; }
void bo_interreduce(void* pl, void* out) { polyvec r = interreduce(PL(pl)); pv_free(PL(out)); *PL(out) = r; }
void bo_buchberger(void* plF, const int* pairs, int npairs, int selection, int elim, int rewards,
                   int sort_input, int sort_reducers, double gamma, int has_seed, int seed,
                   void* out, double* stats) {
  (void)sort_input;
  bstats st; polyvec res;
  if (npairs < 0) {
    res = buchberger_gens(PL(plF), selection, elim, rewards, sort_reducers, gamma, has_seed, seed, out != NULL, &st);
  } else {
    pairvec S; pr_init(&S);
    for (int k = 0; k < npairs; k++) { spair s = {pairs[2 * k], pairs[2 * k + 1]}; pr_push(&S, s); }
    res = buchberger_pairs(PL(plF), &S, selection, elim, rewards, sort_reducers, gamma, has_seed, seed, out != NULL, &st);
    pr_free(&S);
  }
  if (out) { pv_free(PL(out)); *PL(out) = res; }
  stats[0] = st.zero_reductions; stats[1] = st.nonzero_reductions; stats[2] = st.polynomial_additions;
  stats[3] = st.total_reward; stats[4] = st.discounted_return;
}

void bo_cyclic(int n, void* out) { polyvec r = cyclic_ideal(n); pv_free(PL(out)); *PL(out) = r; }
int bo_basis(int n, int d, int* exps, int cap) {
  monovec B = basis(n, d);
  for (int k = 0; k < B.n && k < cap; k++) for (int x = 0; x < BO_N; x++) exps[k * BO_N + x] = B.m[k].e[x];
  int len = B.n;
  free(B.m);
  return len;
}
int bo_degree_distribution(int n, int d, int dist, int constants, double* probs) {
  int* w = (int*)malloc((size_t)(d + 2) * sizeof(int));
  int nw = degree_weights(n, d, dist, constants, w);
  discrete_dist dd = dd_make(w, nw);
  for (int i = 0; i < dd.n; i++) probs[i] = dd.prob[i];
  int len = dd.n;
  dd_free(&dd); free(w);
  return len;
}
void* bo_gen_new(const char* dist) { return parse_ideal_dist(dist); }
void bo_gen_free(void* g) { gen_free((gen*)g); }
void bo_gen_seed(void* g, int seed) { rng_seed(&((gen*)g)->rng, seed); }
int bo_gen_nvars(void* g) { return ((gen*)g)->n; }
void bo_gen_next(void* g, void* out) { gen_next((gen*)g, PL(out)); }
void* bo_gen_copy(void* g) { return gen_clone((gen*)g); }

static env* env_make(gen* g, int elim, int rewards, int sort_input, int sort_reducers) {
  if (!g) return NULL;
  env* e = (env*)calloc(1, sizeof(env));
  e->g = g; e->elim = elim; e->rewards = rewards; e->sort_input = sort_input; e->sort_reducers = sort_reducers;
  pv_init(&e->G); pr_init(&e->P);
  return e;
}
void* bo_env_new(const char* dist, int elim, int rewards, int sort_input, int sort_reducers) {
  return env_make(parse_ideal_dist(dist), elim, rewards, sort_input, sort_reducers);
}
void* bo_env_new_fixed(void* pl, int elim, int rewards, int sort_input, int sort_reducers) {
  return env_make(gen_fixed(PL(pl)), elim, rewards, sort_input, sort_reducers);
}
void bo_env_free(void* p) { env* e = (env*)p; if (!e) return; gen_free(e->g); pv_free(&e->G); pr_free(&e->P); free(e->ord); free(e); }
void* bo_env_copy(void* p) { /* buchberger.cpp:279-297: deep copy incl. generator + RNG state */
  env* s = (env*)p;
  env* e = (env*)calloc(1, sizeof(env));
  *e = *s;
  e->g = gen_clone(s->g);
  e->G = pv_clone(&s->G);
  e->P = pr_clone(&s->P);
  e->ord = (int*)malloc((size_t)(s->nord ? s->nord : 1) * sizeof(int));
  memcpy(e->ord, s->ord, (size_t)s->nord * sizeof(int));
  e->cord = s->nord ? s->nord : 1;
  return e;
}
void bo_env_seed(void* e, int seed) { rng_seed(&((env*)e)->g->rng, seed); }
int bo_env_nvars(void* e) { return ((env*)e)->g->n; }
void bo_env_reset(void* e) { env_reset((env*)e); }
double bo_env_step_pair(void* e, int i, int j) { spair s = {i, j}; return env_step((env*)e, s); }
double bo_env_step(void* p, int action) { env* e = (env*)p; return env_step(e, e->P.p[action]); } /* buchberger.cpp:399 */
double bo_env_value(void* e, const char* strategy, double gamma) { return env_value((env*)e, strategy, gamma); }
int bo_env_nG(void* e) { return ((env*)e)->G.n; }
int bo_env_nP(void* e) { return ((env*)e)->P.n; }
void bo_env_pairs(void* p, int* out) { env* e = (env*)p; for (int k = 0; k < e->P.n; k++) { out[2 * k] = e->P.p[k].i; out[2 * k + 1] = e->P.p[k].j; } }
int bo_env_poly_nterms(void* e, int i) { return ((env*)e)->G.p[i].n; }
int bo_env_poly_sugar(void* e, int i) { return ((env*)e)->G.p[i].sug; }
void bo_env_poly_get(void* e, int i, int* coef, int* exps) { poly_to_flat(&((env*)e)->G.p[i], coef, exps); }
void bo_env_reducer_order(void* p, int* out) { env* e = (env*)p; memcpy(out, e->ord, (size_t)e->nord * sizeof(int)); }
/* buchberger.cpp:354-370 + 391-394/403-406 */
static void lead_monomials_vector(const poly* f, int k, int n, int* out) {
  int i = 0;
  for (; i < f->n && i < k; i++) for (int j = 0; j < n; j++) out[i * n + j] = f->t[i].m.e[j];
  for (; i < k; i++) for (int j = 0; j < n; j++) out[i * n + j] = 0;
}
void bo_env_obs(void* p, int k, int n, int* out) {
  env* e = (env*)p;
  int w = n * k;
  for (int r = 0; r < e->P.n; r++) {
    lead_monomials_vector(&e->G.p[e->P.p[r].i], k, n, out + r * 2 * w);
    lead_monomials_vector(&e->G.p[e->P.p[r].j], k, n, out + r * 2 * w + w);
  }
}
long long bo_env_last_step_bytes(void* e) { return ((env*)e)->last_bytes; }

/* same counter hash as ref_driver.cpp / include/bbx.h (bbx_agent_hash) */
static uint32_t agent_hash(uint32_t seed, uint32_t t) {
  uint64_t z = ((uint64_t)seed << 32 | t) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}
double bo_bench_random(const char* dist, int k, int nenvs, int nsteps, int seed0, int agent_seed0,
                       long long* total_steps, long long* total_additions, unsigned long long* checksum) {
  long long steps = 0, adds = 0;
  unsigned long long cs = 0;
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int ei = 0; ei < nenvs; ei++) {
    env* e = (env*)bo_env_new(dist, ELIM_GM, REW_ADDITIONS, 0, 1);
    if (!e) return -1.0;
    int n = e->g->n;
    int* obs = NULL; int obs_cap = 0;
    bo_env_seed(e, seed0 + ei);
    env_reset(e);
    for (int t = 0; t < nsteps; t++) {
      int action = (int)(((uint64_t)agent_hash((uint32_t)(agent_seed0 + ei), (uint32_t)t) * (uint32_t)e->P.n) >> 32);
      double r = env_step(e, e->P.p[action]);
      int need = e->P.n * 2 * n * k;
      if (need > obs_cap) { obs_cap = need + 256; obs = (int*)realloc(obs, (size_t)obs_cap * sizeof(int)); }
      bo_env_obs(e, k, n, obs); /* LeadMonomialsEnv::step rebuilds the state every step */
      steps++;
      adds += (long long)(-r);
      cs = cs * 1000003ull + (unsigned long long)((long long)need * 31 + (long long)(-r));
      if (e->P.n == 0) env_reset(e);
    }
    free(obs);
    bo_env_free(e);
  }
  clock_gettime(CLOCK_MONOTONIC, &t1);
  *total_steps = steps; *total_additions = adds; *checksum = cs;
  return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* position-keyed commutative word hash (bbx_mix64 of deepgroebner_amd/csrc/bbx_common.h; oracle/trace.py fnv64) */
static unsigned long long mix64w(unsigned long long idx, unsigned int word) {
  unsigned long long z = ((idx << 32) | (unsigned long long)word) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* ONE environment under the counter-hash agent for nsteps steps: what the full-size GPU tests compare every device
 * environment with.  nobs <= 0: observation width of the environment's own variable count.  out8 = steps, additions,
 * algorithmic bytes (observation term included), episodes finished, final |G|, final |P|, hash of the final
 * state as the int32 word stream [per basis element: nterms, (coef, exps[8])...][pairs i, j ...][reducer order]
 * (tests: fnv64 of _state_words), zero reductions.  Returns 0, or -1 for an unknown distribution. */
int bo_run_random(const char* dist, int k, int seed, int agent_seed, int nsteps, int auto_reset, int nobs, long long* out8) {
  env* e = (env*)bo_env_new(dist, ELIM_GM, REW_ADDITIONS, 0, 1);
  if (!e) return -1;
  const int n = nobs > 0 ? nobs : e->g->n;
  long long steps = 0, adds = 0, bytes = 0, episodes = 0, zero = 0;
  bo_env_seed(e, seed);
  env_reset(e);
  for (int t = 0; t < nsteps && e->P.n > 0; t++) {
    const int action = (int)(((uint64_t)agent_hash((uint32_t)agent_seed, (uint32_t)t) * (uint32_t)e->P.n) >> 32);
    const int nG0 = e->G.n;
    const double r = env_step(e, e->P.p[action]);
    steps++; adds += (long long)(-r);
    bytes += e->last_bytes + 4LL * e->P.n * 2 * n * k;
    if (e->G.n == nG0) zero++;
    if (e->P.n == 0) { episodes++; if (auto_reset) env_reset(e); }
  }
  unsigned long long h = 0, at = 0;
  for (int g = 0; g < e->G.n; g++) {
    const poly* f = &e->G.p[g];
    h += mix64w(at++, (unsigned int)f->n);
    for (int t = 0; t < f->n; t++) {
      h += mix64w(at++, (unsigned int)f->t[t].c);
      for (int v = 0; v < BO_N; v++) h += mix64w(at++, (unsigned int)f->t[t].m.e[v]);
    }
  }
  for (int r = 0; r < e->P.n; r++) { h += mix64w(at++, (unsigned int)e->P.p[r].i); h += mix64w(at++, (unsigned int)e->P.p[r].j); }
  for (int r = 0; r < e->nord; r++) h += mix64w(at++, (unsigned int)e->ord[r]);
  out8[0] = steps; out8[1] = adds; out8[2] = bytes; out8[3] = episodes; out8[4] = e->G.n; out8[5] = e->P.n;
  out8[6] = (long long)h; out8[7] = zero;
  bo_env_free(e);
  return 0;
}
