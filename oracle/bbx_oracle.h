/* TEST INFRASTRUCTURE — the parity oracle.  NOT part of the product.
 *
 * A plain-C, single-threaded restatement of the reference's BuchbergerEnv step
 * path (deepgroebner/polynomials.{h,cpp}, buchberger.{h,cpp}, ideals.{h,cpp} of
 * dylanpeifer/deepgroebner).  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it; the product (libbbx.so) never does.
 *
 * Parity status: PINNED — against (1) the known answers of the reference's own
 * test-suite (tests/test_polynomials.cpp, test_buchberger.cpp, test_ideals.cpp,
 * test_buchberger.py, transliterated in tests/), (2) golden traces generated
 * from the compiled reference itself (oracle/_ref, oracle/make_golden.py ->
 * tests/golden/), and (3) live comparison with oracle/_ref when it is present.
 *
 * Flat polynomial exchange format used across this API: a polynomial is
 * (nterms, coef[nterms], exps[nterms*8]) with int32 entries.
 */
#ifndef BBX_ORACLE_H
#define BBX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- polynomial lists -------------------------------------------------- */
void* bo_pl_new(void);
void  bo_pl_free(void* pl);
void  bo_pl_clear(void* pl);
int   bo_pl_len(void* pl);
void  bo_pl_add(void* pl, int nterms, const int* coef, const int* exps); /* Polynomial ctor: sorts terms */
int   bo_pl_nterms(void* pl, int i);
int   bo_pl_sugar(void* pl, int i);
void  bo_pl_get(void* pl, int i, int* coef, int* exps);

/* ---- GF(32003) and monomials ------------------------------------------- */
int bo_coef_norm(int a);
int bo_coef_add(int a, int b);
int bo_coef_sub(int a, int b);
int bo_coef_mul(int a, int b);
int bo_coef_div(int a, int b);
int bo_mono_gt(const int* a, const int* b);

/* ---- polynomial arithmetic, spoly, reduce, update ----------------------- */
void bo_poly_add(void* pl, int i, int j, void* out);
void bo_poly_sub(void* pl, int i, int j, void* out);
void bo_poly_mul(void* pl, int i, int j, void* out);
void bo_parse_polynomial(const char* s, void* out);
void bo_spoly(void* pl, int i, int j, void* out);
int  bo_reduce(void* plg, int gi, void* plF, void* out);
int  bo_update(void* plG, int* pairs, int npairs, void* plf, int fi, int elim);
void bo_minimalize(void* pl, void* out);
void bo_interreduce(void* pl, void* out);
void bo_buchberger(void* plF, const int* pairs, int npairs, int selection, int elim, int rewards,
                   int sort_input, int sort_reducers, double gamma, int has_seed, int seed,
                   void* out, double* stats);

/* ---- ideal generators ---------------------------------------------------- */
void  bo_cyclic(int n, void* out);
int   bo_basis(int n, int d, int* exps, int cap);
int   bo_degree_distribution(int n, int d, int dist, int constants, double* probs);
void* bo_gen_new(const char* dist);
void  bo_gen_free(void* g);
void  bo_gen_seed(void* g, int seed);
int   bo_gen_nvars(void* g);
void  bo_gen_next(void* g, void* out);
void* bo_gen_copy(void* g);

/* ---- BuchbergerEnv -------------------------------------------------------- */
void*  bo_env_new(const char* dist, int elim, int rewards, int sort_input, int sort_reducers);
void*  bo_env_new_fixed(void* pl, int elim, int rewards, int sort_input, int sort_reducers);
void   bo_env_free(void* e);
void*  bo_env_copy(void* e);
void   bo_env_seed(void* e, int seed);
int    bo_env_nvars(void* e);
void   bo_env_reset(void* e);
double bo_env_step_pair(void* e, int i, int j);
double bo_env_step(void* e, int action);
double bo_env_value(void* e, const char* strategy, double gamma);
int    bo_env_nG(void* e);
int    bo_env_nP(void* e);
void   bo_env_pairs(void* e, int* out);
int    bo_env_poly_nterms(void* e, int i);
int    bo_env_poly_sugar(void* e, int i);
void   bo_env_poly_get(void* e, int i, int* coef, int* exps);
void   bo_env_reducer_order(void* e, int* out);
void   bo_env_obs(void* e, int k, int n, int* out);
/* algorithmic bytes of the last step (SURVEY.md section 8d formula) */
long long bo_env_last_step_bytes(void* e);
int bo_stat_max_terms(int reset);   /* longest polynomial held by a reduction of this thread (test statistic) */

/* ---- batch driver used as the CPU baseline ("port") ----------------------- */
double bo_bench_random(const char* dist, int k, int nenvs, int nsteps, int seed0, int agent_seed0,
                       long long* total_steps, long long* total_additions, unsigned long long* checksum);

/* one environment under the counter-hash agent: counters + final-state hash (full-size GPU parity tests) */
int bo_run_random(const char* dist, int k, int seed, int agent_seed, int nsteps, int auto_reset, int nobs, long long* out8);

#ifdef __cplusplus
}
#endif
#endif
