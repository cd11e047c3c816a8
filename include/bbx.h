/* bbx — MI355X-native BuchbergerEnv step path: the C ABI of libbbx.so
 *
 * This is the drop-in boundary for the one hot path of dylanpeifer/deepgroebner that this
 * project replaces: everything the reference's Cython binding (deepgroebner/wrapped.pyx:11-38,
 * declared in deepgroebner/buchberger.pxd:8-18) calls on the C++ class LeadMonomialsEnv
 * (deepgroebner/buchberger.h:224-257, buchberger.cpp:373-408) — generalised from one
 * environment per object to a batch of independent environments per handle, because the device
 * wants thousands of them per launch.  batch == 1 is exactly the reference's single env.
 *
 * Conventions: plain C, caller-allocated outputs, every entry point returns 0 or a negative
 * bbx_status; nothing throws across the ABI; bbx_last_error() (thread-local) explains the last
 * failure.  One host thread per handle.  The library owns all device memory.  It fails loudly
 * (BBX_E_DEVICE) when no HIP device is usable: there is no CPU fallback in the product.
 */
#ifndef BBX_H
#define BBX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bbx_batch bbx_batch;

enum bbx_status {
  BBX_OK = 0,
  BBX_E_ARG = -1,         /* bad argument / unparsable distribution string */
  BBX_E_DEVICE = -2,      /* HIP error or no device */
  BBX_E_CAPACITY = -3,    /* an environment exceeded its configured capacity (see bbx_env_status) */
  BBX_E_GENERATOR = -4,   /* the ideal generator failed (the reference would have thrown) */
  BBX_E_UNSUPPORTED = -5, /* the library has no code for the request (e.g. more than 8 variables, a policy shape or kernel class without a built-in policy) */
  BBX_E_ACTION = -6       /* an action index was outside [0, rows) */
};

/* elimination / reward selectors (reference enums buchberger.h:58 and :93) */
enum { BBX_GEBAUERMOELLER = 0, BBX_LCM = 1, BBX_NONE = 2 };
enum { BBX_ADDITIONS = 0, BBX_REDUCTIONS = 1 };
/* built-in device-side agents for bbx_rollout (0 = actions supplied by the caller) */
enum { BBX_EXTERNAL = 0, BBX_RANDOM_HASH = 1, BBX_DEGREE = 2, BBX_FIRST = 3, BBX_NORMAL = 4, BBX_SUGAR = 5,
       /* the remaining SelectionType values of buchberger.h:111 / buchberger.cpp:200-240: the reversed orders, and
          Random as the reference draws it from a seeded std::default_random_engine (see bbx_seed_strategy) */
       BBX_LAST = 6, BBX_CODEGREE = 7, BBX_STRANGE = 8, BBX_SPICE = 9, BBX_RANDOM_STD = 10 };

/* Per-environment capacities; 0 picks a default suited to the distribution.  Hard limits that remain: 65534 basis
 * elements (pairs are two 16-bit indices), 65535 terms per BASIS element, 2^22 terms per intermediate polynomial. */
typedef struct bbx_caps {
  int32_t max_basis;      /* |G|  (<= 65535) */
  int32_t max_pairs;      /* |P| */
  int32_t arena_terms;    /* total terms of all basis polynomials */
  int32_t max_poly_terms; /* longest intermediate polynomial during spoly/reduce; basis elements hold <= 65535 terms */
  int32_t queue_slots;    /* pre-generated ideals buffered per environment for device-side resets */
  int32_t lds_max_basis;  /* |G| up to which a small (3-variable binomial) environment is kept register/LDS-resident
                             for a whole launch; 0 = default (256 for the class of the reference's C++ LeadMonomialsEnv —
                             Gebauer-Moeller, sorted reducers —, its policy kernels and the other eliminations: 128),
                             negative = never */
  int32_t wide_waves;     /* fixed ideals (long polynomials): waves of the workgroup that serves ONE environment;
                             0 = default (8 when batch <= 4096, else one wave per environment), negative = never */
  int32_t general_class;  /* non-zero: never use the binomial kernel class (term arena + general merges even for
                             binomial ideals); for testing the general path on the same inputs */
  int32_t wide_lds_terms; /* wide class (fixed ideals, one workgroup per environment): terms of the polynomial being
                             reduced kept in LDS (also sizes the reducer window and table); 0 = as much as the LDS
                             holds.  Longer polynomials continue on HBM-resident buffers, so this only affects speed */
  int32_t no_growth;      /* 0 (default): max_basis / max_pairs / arena_terms / max_poly_terms are STARTING sizes — an
                             environment that outgrows one stops before the step that does not fit, the library doubles
                             that array for the whole batch (device memory permitting) and the environment continues, as
                             the reference's heap vectors would (polynomials.h:71-94): capacity costs time, never results.
                             Non-zero: they are hard limits and exceeding one is BBX_E_CAPACITY */
} bbx_caps;

/* One record per environment per step of a traced rollout (parity tests). */
typedef struct bbx_trace_rec {
  int32_t action, rows, basis_size, done;
  double reward;
  uint64_t obs_hash, pairs_hash, newpoly_hash;
} bbx_trace_rec;

/* ---- construction ------------------------------------------------------------------------------
 * Replaces LeadMonomialsEnv::LeadMonomialsEnv(ideal_dist, sort_input, sort_reducers, k)
 * (buchberger.cpp:373-381; reached from wrapped.pyx:14-16) and, for the options the Cython class
 * ignores but the Python class honours, BuchbergerEnv's constructor (buchberger.cpp:269-276;
 * buchberger.py:320-326).  ideal_dist uses the reference grammar (ideals.cpp:103-143). */
int bbx_create(const char* ideal_dist, int elimination, int rewards, int sort_input, int sort_reducers,
               int k, int batch, int device, const bbx_caps* caps, bbx_batch** out);
/* Same over a fixed ideal (FixedIdealGenerator, ideals.h:116-138): polynomial p has nterms[p] terms,
 * coefs/exps (8 ints per term) are concatenated.  nvars_obs <= 0 reproduces the reference's
 * "largest variable index" quirk (ideals.cpp:146-154). */
int bbx_create_fixed(int npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps, int nvars_obs,
                     int elimination, int rewards, int sort_input, int sort_reducers,
                     int k, int batch, int device, const bbx_caps* caps, bbx_batch** out);
/* A list of ideals instead of one (what scripts/make_strat.cpp:12-72 iterates over): environment e is reset to ideals
 * e, e + batch, e + 2*batch, ... (wrapping around).  npolys[nideals]; nterms per polynomial, coefs, exps concatenated.
 * Together with bbx_rollout(agent, huge nsteps, auto_reset = 0) and bbx_stats this yields the reference's
 * ZeroReductions / NonzeroReductions / PolynomialAdditions columns per ideal and strategy. */
int bbx_create_ideals(int nideals, const int32_t* npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps,
                      int nvars_obs, int elimination, int rewards, int sort_input, int sort_reducers,
                      int k, int batch, int device, const bbx_caps* caps, bbx_batch** out);
void bbx_destroy(bbx_batch* b);
/* LeadMonomialsEnv copy constructor (buchberger.pxd:11, wrapped.pyx:35-38): deep clone incl. the
 * generators' RNG state. */
int bbx_copy(const bbx_batch* b, bbx_batch** out);

/* In-batch clones for tree search (the reference's mcts.py:89,96,147 / az.py:82 copy the env per expanded node):
 * environment src[i] overwrites environment dst[i] — device record, queued ideals and generator RNG state — with no
 * allocation, so a batch can serve as a pool of search nodes.  src and dst must not overlap. */
int bbx_clone_envs(bbx_batch* b, int n, const int32_t* src, const int32_t* dst);

/* ---- LeadMonomialsEnv::seed (buchberger.h:243, wrapped.pyx:28-30): one seed per environment ---- */
int bbx_seed(bbx_batch* b, const int64_t* seeds);
/* seeds of the built-in BBX_RANDOM_HASH agent: action = bbx_agent_action(seed, t, rows), t = steps taken so far */
int bbx_seed_agent(bbx_batch* b, const uint32_t* seeds);
/* seeds of the BBX_RANDOM_STD agent: rng.seed(seed) of buchberger(..., SelectionType::Random, ..., seed)
 * (buchberger.cpp:200-203); every pick is choice(P.begin(), P.end(), rng) (buchberger.cpp:244, ideals.h:68-73).
 * scripts/make_strat.cpp:66 passes the same seed for every ideal. */
int bbx_seed_strategy(bbx_batch* b, const int64_t* seeds);

/* ---- LeadMonomialsEnv::reset (buchberger.cpp:384-395, wrapped.pyx:18-21) ----------------------
 * mask == NULL resets every environment, else those with mask[e] != 0.  rows[e] = |P|. */
int bbx_reset(bbx_batch* b, const uint8_t* mask, int32_t* rows);

/* ---- LeadMonomialsEnv::step (buchberger.cpp:398-408, wrapped.pyx:23-26) -----------------------
 * actions[e] indexes the rows of environment e's observation.  Environments that are done are
 * left untouched (reward 0).  rewards/dones/rows may be NULL.
 * Small batches (<= 8 environments) on the class of the reference's C++ LeadMonomialsEnv: the step calls of a loop (from
 * the fifth call in a row on: bbx_step, bbx_step_autoreset, bbx_step_obs) feed one resident kernel through pinned host memory
 * instead of launching one each — a host mailbox session; any other call on the handle ends it first.
 * Results are those of one launch per step; the environment variable BBX_NO_MAILBOX restores that.  Batches of up to 64
 * environments read the actions from and write outputs and observation rows to pinned host memory (no copy calls). */
int bbx_step(bbx_batch* b, const int32_t* actions, double* rewards, uint8_t* dones, int32_t* rows);

/* The vectorised-environment convention: an environment whose episode ends with this step is reset inside the same
 * launch (dones[e] = 1, rows[e] / the next bbx_obs describe the NEW episode), saving the separate bbx_reset call. */
int bbx_step_autoreset(bbx_batch* b, const int32_t* actions, double* rewards, uint8_t* dones, int32_t* rows);

/* nsteps steps per environment in one go with a device-side agent, optionally re-drawing a new
 * ideal whenever an episode ends (what the reference's scripts/random_episodes.cpp:13-29 loop does
 * on the host).  rewards/dones/rows describe the LAST step. */
int bbx_rollout(bbx_batch* b, int agent, int nsteps, int auto_reset, double* rewards, uint8_t* dones, int32_t* rows);

/* ---- the observation: `state` of LeadMonomialsEnv (buchberger.h:247-248; wrapped.pyx:20,25) ---
 * out is int32 [batch, max_rows, cols], cols = 2*nvars*k; rows beyond |P| are filled with -1 when
 * fill != 0 (the padding the reference's agents apply, pg.py:217-226). */
int bbx_obs(bbx_batch* b, int32_t* out, int max_rows, int fill);
/* One call per vector step of the Gym-style loop: the step (actions != NULL; NULL = only refresh the observation), its
 * outputs, and the observation of every environment as a RAGGED block: *offsets -> int32 [batch + 1] row offsets,
 * *obs -> int32 [offsets[batch], cols], environment e owning rows offsets[e] .. offsets[e+1]-1 (exactly the `state`
 * matrices wrapped.pyx:20,25 returns, back to back, no padding).  Both point into pinned memory owned by the handle and
 * stay valid until the next call on it.  One kernel launch for the step, two small ones to pack, two device-to-host
 * copies sized by what the pair sets actually hold (3-4 MB instead of a 25 MB padded block at batch 4096). */
int bbx_step_obs(bbx_batch* b, const int32_t* actions, int auto_reset, double* rewards, uint8_t* dones, int32_t* rows,
                 const int32_t** obs, const int32_t** offsets);
int bbx_cols(const bbx_batch* b);
int bbx_nvars(const bbx_batch* b);
int bbx_batch_size(const bbx_batch* b);

/* ---- LeadMonomialsEnv::value (buchberger.h:245, buchberger.cpp:332-351, wrapped.pyx:32-33) ----
 * discounted return of a full Buchberger rollout from environment idx's current state.
 * Unknown strategies select First, as the reference's std::map lookup does. */
int bbx_value(bbx_batch* b, int idx, const char* strategy, double gamma, double* out);
/* the same for every environment of the batch at once: out[batch].  "random" and "sample" roll out under
 * buchberger(..., SelectionType::Random, ..., seed) (buchberger.cpp:200-203, 244: a std::default_random_engine seeded per
 * rollout, every pick choice(P.begin(), P.end(), rng)); the reference seeds each rollout from std::random_device, here the
 * seeds come from a stream of the handle's own (started from its seed base: reproducible under BBX_DEFAULT_SEED). */
int bbx_values(bbx_batch* b, const char* strategy, double gamma, double* out);
/* ... with the seeds of the Random rollouts given: seeds[batch] for "random", seeds[batch][100] for "sample" (the 100
 * Random rollouts of environment e; its Degree rollout needs none); other strategies ignore them.  Equal to the
 * reference's buchberger(G, P, Random, ..., seed).discounted_return per rollout, bit for bit. */
int bbx_values_seeded(bbx_batch* b, const char* strategy, double gamma, const int64_t* seeds, double* out);

/* ---- same calls on caller-owned DEVICE buffers (e.g. torch tensors), asynchronous on `stream` ----
 * (hipStream_t passed as void*; NULL = the default stream).  obs may be NULL.
 * obs_fill: 0 = rows beyond |P| are left alone; 1 = they are padded with -1 (pg.py:217-226); 2 = incremental padding:
 * the caller vouches that d_obs and d_rows still hold what the previous call on this handle wrote, so only the rows
 * that stopped being valid are re-padded (classes without the incremental path pad everything, like 1).
 * Any number of these calls may be queued before a bbx_sync.  An environment that outgrows a capacity stops at that step:
 * with one step between two waits bbx_sync enlarges the records and takes the step (nothing to see); with several, the
 * environment has sat out the later ones and d_actions meanwhile holds a later step's actions, so bbx_sync enlarges the
 * records and returns BBX_E_CAPACITY naming the environment — its steps of that chain are missing, later calls proceed. */
int bbx_step_device(bbx_batch* b, const int32_t* d_actions, double* d_rewards, uint8_t* d_dones, int32_t* d_rows,
                    int32_t* d_obs, int obs_rows, int obs_fill, void* stream);
/* the same with the vectorised-environment convention of bbx_step_autoreset (finished episodes restart inside the call) */
int bbx_step_device_autoreset(bbx_batch* b, const int32_t* d_actions, double* d_rewards, uint8_t* d_dones, int32_t* d_rows,
                              int32_t* d_obs, int obs_rows, int obs_fill, void* stream);
/* ---- the consumer of the padded observation block: the reference's default policy network on the device ----------
 * ParallelMultilayerPerceptron([hidden]) (networks.py:522-571: ParallelEmbeddingLayer :49-95 with one dense relu layer,
 * ParallelDecidingLayer :414-460) evaluated on d_obs [batch, obs_rows, cols] over the d_rows[e] valid rows of each
 * environment (the -1 padding is masked out there, plays no part here), log-softmax over the rows and ONE action drawn
 * by inverse CDF from the uniform number d_u[e] in [0, 1) — the step of pg.py:451-503's run_episode that used to cross to
 * the host every step.  fp32, the hidden layer on the matrix cores (exact f32 MFMA); cols <= 64, hidden <= 256, at most
 * 2048 rows per environment (obs_rows > 2048: BBX_E_UNSUPPORTED; a policy rollout without an observation block whose pair
 * set outgrows 2048 rows: BBX_E_CAPACITY from bbx_sync — never a silent cut).  exp / log of the softmax are the
 * hardware's fast forms (__expf / __logf): log-probabilities agree with an IEEE evaluation to ~2e-4 (tests/test_rollout.py).
 * The weights are handed over PREPARED: bbx_pmlp_prepare copies d_w1 [cols][hidden] (the layout of
 * torch.nn.Linear(...).weight.t()), d_b1 [hidden], d_w2 [hidden], b2 into d_prepared (bbx_pmlp_prepared_floats(cols,
 * hidden) floats, 16-byte aligned) zero-padded to the kernels' tile sizes; call it again whenever the weights change.
 * d_actions[e] in [0, rows), d_logprobs[e] = its log-probability. */
int bbx_pmlp_prepared_floats(int cols, int hidden);        /* < 0: shape not supported */
int bbx_pmlp_prepare(const float* d_w1, const float* d_b1, const float* d_w2, float b2, int cols, int hidden, float* d_prepared, void* stream);
int bbx_pmlp_act(const int32_t* d_obs, const int32_t* d_rows, int batch, int obs_rows, int cols, const float* d_prepared, int hidden,
                 const float* d_u, int32_t* d_actions, float* d_logprobs, void* stream);
/* The same for ParallelMultilayerPerceptron(hidden_layers=[hidden1, hidden2]) (networks.py:562-571: two dense layers in the
 * embedding): logit_r = w3 . relu(W2^T relu(W1^T x_r + b1) + b2) + b3, both layers on the matrix cores in exact f32, the
 * second layer's weights staged in LDS once per workgroup (bbx_pmlp2.hip); cols <= 64, hidden1, hidden2 <= 128, at most 2048
 * rows per environment.  d_w1 [cols][hidden1], d_b1 [hidden1], d_w2 [hidden1][hidden2], d_b2 [hidden2], d_w3 [hidden2],
 * d_b3 [1] (the transposed layouts of torch.nn.Linear weights; every argument on the device: preparing never reads back).
 * Stands alone in front of bbx_step_device_autoreset (two launches per vector step, both recordable into a HIP graph);
 * the fused per-step / rollout / session forms exist for one hidden layer only. */
int bbx_pmlp2_prepared_floats(int cols, int hidden1, int hidden2);   /* < 0: shape not supported */
int bbx_pmlp2_prepare(const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2, const float* d_w3, const float* d_b3,
                      int cols, int hidden1, int hidden2, float* d_prepared, void* stream);
int bbx_pmlp2_act(const int32_t* d_obs, const int32_t* d_rows, int batch, int obs_rows, int cols, const float* d_prepared, int hidden1, int hidden2,
                  const float* d_u, int32_t* d_actions, float* d_logprobs, void* stream);
/* ... and for three hidden layers (hidden_layers=[hidden1, hidden2, hidden3], each <= 128): the same kernel with a middle layer
 * whose output stays in registers as the next layer's B operands; d_w3 [hidden2][hidden3], d_b3 [hidden3], d_w4 [hidden3],
 * d_b4 [1].  All three layers are padded to the widest one's tile size (64 or 128 units). */
int bbx_pmlp3_prepared_floats(int cols, int hidden1, int hidden2, int hidden3);   /* < 0: shape not supported */
int bbx_pmlp3_prepare(const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2, const float* d_w3, const float* d_b3,
                      const float* d_w4, const float* d_b4, int cols, int hidden1, int hidden2, int hidden3, float* d_prepared, void* stream);
int bbx_pmlp3_act(const int32_t* d_obs, const int32_t* d_rows, int batch, int obs_rows, int cols, const float* d_prepared, int hidden1, int hidden2,
                  int hidden3, const float* d_u, int32_t* d_actions, float* d_logprobs, void* stream);
/* One vector step with the policy in the loop: bbx_pmlp_act on the block the previous call left in d_obs / d_rows, then
 * bbx_step_device_autoreset with the sampled rows as actions, which rewrites d_obs / d_rows (the inner loop of
 * pg.py:451-503 run_episode, batched).  Where the step kernel has the policy built in (the register/LDS-resident class
 * with accounting off, 33..128 hidden units) this is ONE kernel launch; otherwise the two calls it stands for.
 * d_actions and d_logprobs receive what was sampled; arguments as in those two calls. */
int bbx_policy_step_device(bbx_batch* b, const float* d_prepared, int hidden, const float* d_u, int32_t* d_actions, float* d_logprobs,
                           double* d_rewards, uint8_t* d_dones, int32_t* d_rows, int32_t* d_obs, int obs_rows, int obs_fill, void* stream);
/* nsteps vector steps in ONE launch with the policy inside the step kernel (pg.py:451-503 run_episode for a whole batch, no
 * host in the loop and no waiting between environments): at step t every environment writes the observation the policy
 * is about to see to d_obs + t * obs_step_stride (int32 elements; 0 = a single block overwritten every step; rows beyond
 * d_rows[t][e] are not written) and its row count to d_rows[t][e], evaluates the policy, draws the action with d_u[t][e]
 * -> d_actions[t][e], d_logprobs[t][e], takes the step -> d_rewards[t][e], d_dones[t][e]; finished episodes restart
 * (auto-reset).  All per-step arrays are [nsteps][batch]; d_rewards / d_dones / d_rows / d_obs may be null.  Built into the
 * binomial kernel classes (binomial ideals in up to 7 variables, 2nk <= 12 observation columns — <= 20 with more than 3
 * variables —, accounting off, 33..128 hidden units); other batches: BBX_E_UNSUPPORTED (use bbx_policy_step_device).
 * 3 variables with k = 2 run in the register/LDS-resident kernel; environments that outgrow it inside the rollout
 * (|G| > 128 or |P| > 256) are continued, policy included, by the HBM-resident kernel launched right behind, which is
 * also the rollout kernel of every other admitted shape.
 * Batches whose ideals come from the host-side queue (BBX_HOST_GEN, sort_input with more than 16 generators) must hold
 * enough queued ideals for the episodes that end inside the launch (bbx_caps.queue_slots, bbx_prefetch): the host cannot
 * refill in the middle of a launch, and an environment left waiting makes bbx_sync report BBX_E_CAPACITY.
 * Asynchronous like bbx_rollout_device. */
int bbx_policy_rollout_device(bbx_batch* b, const float* d_prepared, int hidden, int nsteps, const float* d_u, int32_t* d_actions,
                              float* d_logprobs, double* d_rewards, uint8_t* d_dones, int32_t* d_rows, int32_t* d_obs, int obs_rows,
                              long long obs_step_stride, void* stream);
/* obs_every_step != 0 materialises the observation in d_obs after every step (what a device-side policy
 * would consume), otherwise only the state at the end of the rollout is written */
int bbx_rollout_device(bbx_batch* b, int agent, int nsteps, int auto_reset, double* d_rewards, uint8_t* d_dones,
                       int32_t* d_rows, int32_t* d_obs, int obs_rows, int obs_fill, int obs_every_step, void* stream);
/* after asynchronous rollouts: waits, refills the ideal queues, finishes environments that had to
 * wait for ideals; returns BBX_E_CAPACITY etc. if any environment failed */
int bbx_sync(bbx_batch* b);

/* ---- persistent sessions ---------------------------------------------------------------------------------------------
 * A launch of K steps ends with its slowest environment (an episode reset costs as much as several steps), so K-step
 * rollouts queued one kernel each leave most of the device idle at the end of every launch: 115 us per 20-step launch for
 * 68 us of work.  With bbx_persistent(b, 1), the first asynchronous bbx_rollout_device call on the register/LDS-resident
 * class (3-variable binomial distributions drawn on the device, Gebauer-Moeller, sorted reducers; built-in agent,
 * auto-reset, lean, untraced, batch <= 4096) starts ONE kernel whose waves keep their environments, and every further
 * call with the same arguments only raises the step total in a device-visible control word; a wave that has taken every
 * step issued so far looks at the word and carries on — environments never wait for each other between calls.  The
 * session ends (its kernel is told to stop after the steps issued, a closing launch takes whatever an environment still
 * owes) with bbx_sync, bbx_join or any other call on the handle; results are those of the same calls as separate
 * launches, bit for bit.  Outputs (rewards / dones / rows / observation block: those of the LAST step) are ready for the
 * caller's stream after bbx_join(b, stream) — device-side: `stream` waits, the host does not — or after bbx_sync.
 * A wave that sees no news for 20 ms leaves by itself (a session never outlives an idle host by more than that). */
int bbx_persistent(bbx_batch* b, int enable);
int bbx_join(bbx_batch* b, void* stream);
/* out5 = {sessions begun, calls that joined a running session, env-steps taken by later kernels of sessions, kernels,
 *         environments that left the register/LDS-resident class (basis beyond 256 elements / 512 pairs) and were continued
 *         by the HBM-resident pass: counted while sessions are enabled} */
int bbx_session_stats(bbx_batch* b, int64_t* out5);
/* Step / reset / observation kernels launched for the handle so far (the kernels of one call: the class's kernel, the
 * continuation pass behind it where the class has one, the second kernel of a wide-class launch of more workgroups than CUs):
 * what a call costs in launches, for tests and for callers who budget them.  No reference counterpart. */
int bbx_kernels_launched(bbx_batch* b, int64_t* out);

/* ---- HIP graphs ---------------------------------------------------------------------------------------------------------
 * The asynchronous device calls (bbx_step_device[_autoreset], bbx_rollout_device, bbx_policy_step_device,
 * bbx_policy_rollout_device) on a batch whose ideals are drawn on the device (or a fixed ideal) only enqueue kernels whose
 * arguments do not depend on host-side counters, so they may be recorded while `stream` is capturing
 * (hipStreamBeginCapture / torch.cuda.graph) together with whatever produces the actions — e.g. a policy of any depth as
 * library GEMMs — and replayed as one graph launch per vector step: the reference's loop `action = policy(state);
 * state, reward, done, _ = env.step(action)` (pg.py:451-465) without a host call per operation.  Launches that would need
 * the host (ideals drawn on the host, persistent sessions, kernel timing) return BBX_E_UNSUPPORTED while capturing.
 * Replays bypass the library, so tell it before the next bbx_sync: bbx_graph_replayed(b, stream) marks the handle as
 * having work in flight on `stream` (the one the graph was replayed on); bbx_sync then waits for it, reports what the
 * replayed steps reported (BBX_E_ACTION, capacities, ...) and continues environments that had to stop, exactly as after
 * the same calls made directly.  One thing invalidates a recording: records enlarged by bbx_sync (an environment outgrew a
 * capacity) live at a new address; the old arrays are kept so that further replays stay harmless, and the next
 * bbx_graph_replayed returns BBX_E_CAPACITY — the steps replayed since did not reach the batch; record the step again. */
int bbx_graph_replayed(bbx_batch* b, void* stream);

/* Algorithmic-byte accounting (stats column 6, the roofline numerator) is on by default; the hand-tuned kernel
 * has a leaner variant without it, selected by bbx_accounting(b, 0).  The count is a property of the workload:
 * bench.py times the lean variant and takes the bytes from an accounting run over a bbx_copy of the same batch. */
int bbx_accounting(bbx_batch* b, int enable);

/* Tops every environment's ring of pre-generated ideals up to queue_slots and uploads them, so that the
 * following rollouts find their inputs resident in HBM (launches themselves only refill EMPTY rings). */
int bbx_prefetch(bbx_batch* b);

/* HIP-event timing of the step-kernel launches on their own stream: returns the milliseconds and launch
 * count accumulated since the previous call, then enables/disables further collection */
int bbx_timing(bbx_batch* b, int enable, double* kernel_ms, int32_t* launches);

/* ---- introspection (tests, checkpoints) -------------------------------------------------------- */
/* per environment 8 values: total_steps, total_additions, episodes, zero_reductions, status,
 * ideals_consumed, algorithmic_bytes (the roofline numerator), basis_size */
int bbx_stats(bbx_batch* b, int64_t* out8);
int bbx_env_status(bbx_batch* b, int32_t* status);
/* the current per-environment capacities (they grow on demand, see bbx_caps.no_growth) and how often they grew:
 * out5 = {max_basis, max_pairs, arena_terms, max_poly_terms, times enlarged} */
int bbx_capacities(bbx_batch* b, int32_t* out5);
int bbx_state_sizes(bbx_batch* b, int idx, int32_t* basis_size, int32_t* npairs, int32_t* nterms_total);
/* G[0..basis_size): nterms[i], then concatenated coefs and exps (8 ints per term); pairs as (i,j);
 * order[r] = index into G of the r-th reducer */
int bbx_state_get(bbx_batch* b, int idx, int32_t* nterms, int32_t* coefs, int32_t* exps, int32_t* pairs, int32_t* order);
/* interreduce(minimalize(G)) of environment idx's basis — the reduced Groebner basis buchberger() returns
 * (buchberger.cpp:102-122, 265) once the environment's pair set is empty; computed on the device (bbx_alg_*).  Call with
 * nterms == NULL for the sizes, then with buffers (nterms[basis_size], coefs/exps[nterms_total(*8)]). */
int bbx_reduced_basis(bbx_batch* b, int idx, int32_t* basis_size, int32_t* nterms_total, int32_t* nterms, int32_t* coefs, int32_t* exps);
int bbx_trace_enable(bbx_batch* b, int capacity_steps);  /* 0 disables */
int bbx_trace_read(bbx_batch* b, int env, int first, int count, bbx_trace_rec* out);

/* ---- ideal generators on their own (reference deepgroebner/ideals.{h,cpp}; host only) ---------- */
typedef struct bbx_gen bbx_gen;
int bbx_gen_create(const char* ideal_dist, bbx_gen** out);
void bbx_gen_destroy(bbx_gen* g);
int bbx_gen_seed(bbx_gen* g, int64_t seed);
int bbx_gen_nvars(const bbx_gen* g);
/* bbx_gen_next draws the next ideal (IdealGenerator::next) and reports its size; bbx_gen_get copies
 * the ideal drawn last: nterms[npolys], coefs[nterms_total], exps[nterms_total*8]; sugars may be NULL */
int bbx_gen_next(bbx_gen* g, int32_t* npolys, int32_t* nterms_total);
int bbx_gen_get(const bbx_gen* g, int32_t* nterms, int32_t* coefs, int32_t* exps, int32_t* sugars);

/* ---- ideals as text (data/stats/<dist>/<dist>.csv lines) ------------------------------------------
 * bbx_parse_ideal reads polynomials such as "413*a^2*b^5*c+32*d^2-5" joined by '|' (parse_polynomial,
 * polynomials.cpp:226-300; parse_ideal_string, scripts/make_strat.cpp:12-19) into the flat layout bbx_create_ideals
 * takes.  It always reports the sizes; with nterms == coefs == exps == NULL it only does that.  Inputs on which the
 * reference has undefined behaviour (variable beyond 'h', zero coefficients, empty polynomials) are BBX_E_ARG.
 * bbx_format_ideal writes the same format (coefficients as signed representatives, the way Macaulay2's make_dist.m2
 * prints them) and returns the length needed excluding the terminator; nothing is written when cap is too small. */
int bbx_parse_ideal(const char* text, int32_t cap_polys, int32_t cap_terms, int32_t* npolys, int32_t* nterms_total,
                    int32_t* nterms, int32_t* coefs, int32_t* exps);
int bbx_format_ideal(int npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps, char* out, int cap);

/* ---- the reference's free functions on device-resident lists of polynomials -----------------------------------------
 * spoly / reduce / update (buchberger.cpp:18-99; buchberger.py:11-147), minimalize / interreduce (buchberger.cpp:102-122;
 * buchberger.py:150-166) and Polynomial +, -, * (polynomials.cpp:148-210), batched: a bbx_alg holds `nlists` independent
 * std::vector<Polynomial>s in HBM (flat input like bbx_create_ideals: npolys[nlists], nterms per polynomial, coefs, exps
 * with 8 ints per term; a polynomial may have 0 terms), every call runs one kernel over all lists (one wavefront each) and
 *   - binop / reduce APPEND their result to every list (G.push_back): read it back with bbx_alg_sizes + bbx_alg_get;
 *   - update treats the LAST element of every list as f and the elements before it as G, and rewrites the pair sets;
 *   - minimalize / interreduce REPLACE every list by the result.
 * Records grow on demand.  GF(32003), grevlex, up to 8 variables, polynomials of at most 65535 terms. */
typedef struct bbx_alg bbx_alg;
int bbx_alg_create(int device, int nlists, const int32_t* npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps, bbx_alg** out);
void bbx_alg_destroy(bbx_alg* a);
/* the bases of n environments of a batch (envs == NULL: environments 0..n-1) as n polynomial lists, copied on the device:
 * with bbx_alg_minimalize + bbx_alg_interreduce the reduced Groebner bases of a whole batch of finished runs
 * (scripts/make_strat.cpp of the reference prints their sizes) without the polynomials visiting the host */
int bbx_alg_from_envs(bbx_batch* b, int n, const int32_t* envs, bbx_alg** out);
/* op: 0 = f + g, 1 = f - g, 2 = f * g, 3 = spoly(f, g); ij[nlists][2] = the elements f, g of every list */
int bbx_alg_binop(bbx_alg* a, int op, const int32_t* ij);
/* reduce(g, F) (buchberger.cpp:24-49): dividend_and_ndivisors[nlists][2] = {index of g, n: F = the list's elements 0..n-1 in list
 * order}; the remainder is appended, steps[nlists] (may be NULL) = successful reductions */
int bbx_alg_reduce(bbx_alg* a, const int32_t* dividend_and_ndivisors, int32_t* steps);
/* update(G, P, f, elimination) (buchberger.cpp:52-99): npairs[nlists] + concatenated pairs (i, j) in; npairs_out[nlists] +
 * pairs_out (room for pairs_cap pairs; may be NULL) = the new pair sets in the reference's order */
int bbx_alg_update(bbx_alg* a, int elimination, const int32_t* npairs, const int32_t* pairs, int32_t* npairs_out, int32_t* pairs_out, int pairs_cap);
int bbx_alg_minimalize(bbx_alg* a);
int bbx_alg_interreduce(bbx_alg* a);
int bbx_alg_sizes(bbx_alg* a, int32_t* npolys, int32_t* nterms_total);          /* per list */
int bbx_alg_get(bbx_alg* a, int list, int32_t* nterms, int32_t* coefs, int32_t* exps, int32_t* sugars);   /* sugars may be NULL */

uint32_t bbx_agent_hash(uint32_t seed, uint32_t t);
uint32_t bbx_agent_action(uint32_t seed, uint32_t t, uint32_t rows);   /* (hash * rows) >> 32 */
const char* bbx_last_error(void);
const char* bbx_version(void);

#ifdef __cplusplus
}
#endif
#endif
