// Shared between the host side of libbbx (bbx_api.cpp) and the HIP kernels (bbx_*.hip, bbx_device.h).
//
// HBM layout
// ----------
// A batch is B independent Buchberger environments.  Each environment owns one
// contiguous *record* in HBM (rec_bytes, 256-B aligned) holding everything the
// step path touches; records never reference each other, so a batch shards
// across wavefronts / GPUs with no exchange.  Inside a record every array is
// 16-B aligned so monomials can be moved with 8-/16-byte loads:
//
//   hdr     BbxHdr                      counters, sizes, status
//   lm      Mono[maxG]                  lead monomial of G[i]      (G order; pair criteria, lcm)
//   slm     Mono[maxG]                  lead monomials in REDUCER order (ascending grevlex; the
//                                       first-divisor scan reads this coalesced, lane k <- slm[k])
//   lcm     Mono[maxG]                  scratch: lcm(LM G[i], LM f) during the Gebauer-Moeller update
//   am      Mono[arena]                 term arena: monomials of all basis polynomials, append-only
//   hm      Mono[5*maxT]                scratch polynomials: h ping/pong, r, merge staging (2*maxT)
//   poff    u32[maxG]                   arena offset of G[i]
//   pairs   u32[maxP]                   pair set P in reference order, i | j<<16
//   sidx    u16[maxG]                   reducer r -> index into G
//   plen    u16[maxG]                   #terms of G[i]       psug u16[maxG] sugar   pinv u16[maxG] 1/LC
//   ac      u16[arena]                  arena coefficients   hc u16[5*maxT] scratch coefficients
//   cp      u8[maxG]                    scratch: G[i] coprime to f
//
// Mono is W 32-bit words of packed u16 pairs (v_pk_max_u16 / v_pk_add_u16 / v_pk_sub_u16 clamp
// operate on it directly): slot s (= variable s) lives in word s/2, half s%2; the LAST slot holds
// the total degree.  W=2 serves n<=3 variables (8 B / monomial), W=4 serves n<=7 (16 B).
#pragma once
#include <stdint.h>

#define BBX_P 32003u
#define BBX_POLICY_MAX_ROWS 2048    /* rows (pairs) per environment the policy kernels score: their logits live in LDS */
#define BBX_MAXVARS 8

// per-environment status (sticky except STARVED)
enum {
  BBX_ST_OK = 0,
  BBX_ST_G_FULL = 1,        // basis capacity exceeded
  BBX_ST_P_FULL = 2,        // pair capacity exceeded
  BBX_ST_ARENA_FULL = 3,    // term arena exhausted
  BBX_ST_POLY_TOO_LONG = 4, // an intermediate polynomial exceeded maxT terms
  BBX_ST_DEG_OVERFLOW = 5,  // total degree above 65535
  BBX_ST_STARVED = 6,       // ideal queue empty: the environment waits for the host to refill
  BBX_ST_BAD_ACTION = 7,    // action index outside [0, |P|)
  BBX_ST_RUNAWAY = 9,       // a reduction exceeded 2^24 rounds (corrupt state guard; never seen in practice)
  BBX_ST_GEN_FAIL = 10,     // the ideal generator failed (no two distinct monomials after 1000 trials: the reference throws)
  BBX_ST_GEN_ZERO = 11,     // a random polynomial cancelled to zero (undefined in the reference)
  BBX_ST_POLY_LIMIT = 12,   // a basis element would have more than 65535 terms (plen[] is 16 bits): a hard limit
  BBX_ST_TIMESLICE = 14,    // transient: a kernel of a persistent session ended its time slice with steps still owed (a kernel
                            // that runs longer than ~100 ms is clocked down to half speed; sessions run in slices of 40 ms)
  BBX_ST_SPILL = 8,         // transient: the state outgrew the LDS-resident class; the HBM-resident pass of the
                            // same launch sequence continues this environment
};

// Capacity statuses are not failures: the kernels leave the record consistent (the step that did not fit has not been
// started: nothing persistent is modified before its last capacity check), the host enlarges the record layout
// (bbx_api.cpp grow_records) and the environment continues.  A launch that finds an environment waiting like that adds
// its steps to the environment's budget instead of replacing it, so no step is lost across asynchronous launches.
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
int bbx_st_capacity(int st) { return st == BBX_ST_G_FULL || st == BBX_ST_P_FULL || st == BBX_ST_ARENA_FULL || st == BBX_ST_POLY_TOO_LONG; }

// the status word of the `lite` block ({status, q_head, budget, |P|} per environment, polled by the host after a
// launch) carries this flag on top of the status code: BbxHdr.obs_trunc != 0
#define BBX_LITE_OBS_TRUNC 0x10000

struct BbxHdr {             // 128 bytes
  int32_t nG, nP, arena_used, status;
  int32_t need_reset, q_head, t, episode_steps;
  int64_t total_steps, total_additions;
  int32_t episodes, zero_reductions;
  uint32_t agent_seed;
  int32_t steps_done;       // steps completed in the last launch
  int32_t budget;           // steps still owed in the current rollout (survives queue starvation)
  int32_t rollout_pos;      // steps completed in the current rollout (trace slot)
  int32_t done_last;        // the last executed step ended an episode
  uint32_t std_rng;         // state of the minstd_rand0 engine behind the reference's seeded Random selection (buchberger.cpp:200-206)
  int64_t alg_bytes;        // algorithmic bytes moved so far (SURVEY.md 8d formula), for the roofline figure
  double vret, vdisc;       // value() rollouts: discounted return so far and the current discount (buchberger.cpp:248-252)
  uint32_t gen_rng;         // state of this environment's ideal generator engine (minstd_rand0) when ideals are drawn
                            // on the device (BbxParams.gen != null); the host-side generators then stay unused
  int32_t obs_trunc;        // != 0: during the current rollout an observation had more rows than the caller's block holds
                            // (rows beyond obs_rows were not written); reported by bbx_sync as BBX_E_CAPACITY
  int32_t sess_done;        // persistent sessions (bbx_persistent): steps of the session's total this environment has taken
  int32_t reserved[3];
};

struct BbxLayout {
  uint32_t W;               // words per monomial (2 or 4)
  uint32_t maxG, maxP, arena, maxT;
  uint32_t off_lm, off_slm, off_lcm, off_am, off_hm, off_poff, off_pairs;
  uint32_t off_sidx, off_plen, off_psug, off_pinv, off_ac, off_hc, off_cp;
  uint32_t rec_bytes;
  // binomial class (kind == 1): every basis polynomial has <= 2 terms, so there is no arena; the
  // record holds, in BASIS order, lm[] / tm[] (lead and tail monomial) and ginfo[] = {lc | tc<<16,
  // 1/lc | sugar<<16}, and in REDUCER order slm[] / stm[] and sinfo[] = {tc | (-tc/lc mod p)<<16, sugar | g<<16},
  // so that one reduction round needs the scan of slm[] plus two independent loads.  tc == 0: no tail.
  uint32_t kind;
  uint32_t off_tm, off_stm, off_ginfo, off_sinfo;
};

// Ideal queue: per environment a ring of `nslots` ideals (or ONE shared slot when fixed != 0).
// Slot words: [npolys, then per polynomial: nterms, sugar, then nterms x (coef, mono W words)].
struct BbxQueue {
  const uint32_t* words;
  uint32_t env_stride;      // words per environment (0 when fixed)
  uint32_t slot_words;
  uint32_t nslots;
  uint32_t fixed;
  uint32_t no_redraw;       // ideal lists (bbx_create_ideals): an ideal whose pair set starts empty IS the episode (buchberger() of it
                            // returns at once, make_strat.cpp:62-66); BuchbergerEnv::reset would draw another (buchberger.cpp:313-314)
  const int32_t* tail;      // [B] ideals produced so far per environment (host-written)
};

struct BbxTraceRec {        // one per environment per step when tracing (tests / parity)
  int32_t action, nP, nG, done;
  double reward;
  uint64_t obs_hash, pairs_hash, newpoly_hash;
};

enum { BBX_AGENT_EXTERNAL = 0, BBX_AGENT_HASH = 1, BBX_AGENT_DEGREE = 2, BBX_AGENT_FIRST = 3, BBX_AGENT_NORMAL = 4, BBX_AGENT_SUGAR = 5,
       // the reversed orders of buchberger.cpp:207-240 and the seeded std::default_random_engine choice of :200-206,244
       BBX_AGENT_LAST = 6, BBX_AGENT_CODEGREE = 7, BBX_AGENT_STRANGE = 8, BBX_AGENT_SPICE = 9, BBX_AGENT_STDRANDOM = 10 };
enum { BBX_ELIM_GM = 0, BBX_ELIM_LCM = 1, BBX_ELIM_NONE = 2 };
enum { BBX_REW_ADDITIONS = 0, BBX_REW_REDUCTIONS = 1 };

// one-hidden-layer PMLP policy (networks.py:49-95, 414-460) for the policy + step launch: device pointers
struct BbxPolicy {
  const float* wp; int32_t hidden;     // prepared weights (bbx_pmlp_prepare) of a [cols] -> [hidden] -> 1 network
  const float* u;                      // [B] uniforms in [0, 1) for the inverse-CDF draw
  int32_t* actions; float* logprobs;   // [B] outputs: the sampled row and its log-probability
  // policy ROLLOUT (nsteps > 1 inside one launch): the arrays above are [nsteps][B] and so are these; the observation
  // the policy saw at step t goes to obs + t * obs_tstride (0: one block, overwritten every step)
  int32_t rollout;                     // 0: per-step call; 1: rollout, register/LDS-resident kernel first; 2: rollout, HBM-resident kernel only
  double* rewards_t; uint8_t* dones_t; int32_t* rows_t; long long obs_tstride;
  // calls of bbx_policy_step_device served by a persistent session: u is [steps][B] (call t reads slice t), every other
  // array is the [B] array of the calls, rewritten at each step (stride_out = 0), and the observation block and row
  // counts describe the state AFTER the step, as that call leaves them (post_obs = 1).  Rollouts: stride_out = B, post_obs = 0.
  int32_t stride_out, post_obs;
};
struct BbxParams {
  char* recs;
  BbxLayout L;              // layout of the records in HBM
  BbxLayout LL;             // layout of the LDS-resident working copy (staged kernel only)
  int32_t fast_G, fast_P;   // capacities of the register/LDS-resident class (bbx_fast.h; its kernels clamp them to what they hold)
  BbxQueue q;
  int32_t B;
  int32_t nsteps;
  int32_t set_budget;       // 1: start a rollout of nsteps steps; 0: continue the one in progress
  int32_t agent;
  int32_t auto_reset;
  int32_t elim, rewards_mode, sort_reducers, k, nvars;
  const struct BbxPolicy* policy;   // HOST pointer or null: a PMLP policy evaluated inside the step kernel (fast class only)
  int32_t sort_input;       // device-drawn ideals: the generators of a new ideal enter in ascending lead-monomial order
                            // (BuchbergerEnv::reset, buchberger.cpp:299-303; at most 16 generators: see gen_sorted_rank)
  const int32_t* actions;   // [B], agent == EXTERNAL
  double* rewards;          // [B] reward of the last executed step
  uint8_t* dones;           // [B]
  int32_t* rows;            // [B] |P| after the step
  int32_t* obs;             // [B, obs_rows, 2*n*k] int32 or null
  int32_t obs_rows;         // row capacity per environment in obs
  int32_t obs_fill;         // 1: pad rows [nP, obs_rows) with -1
  int32_t obs_every_step;   // 1: materialise the observation after every step (what a policy consumes),
                            // 0: only for the state the caller sees when the launch ends
  int32_t value_mode;       // 1: accumulate the discounted return of the rollout (value(), buchberger.cpp:332-351)
  double gamma;
  double* values;           // [B] discounted return when value_mode
  int32_t accounting;       // 1: count algorithmic bytes per step (BbxHdr.alg_bytes); the lean fast kernel omits it
  int32_t pass;             // 0: primary launch; 1: follow-up launch serving only environments with work left
  const uint16_t* inv_table; // [32003] inverses in GF(32003) (L2-resident), binomial class
  const uint32_t* gen;      // device-side generator table (BBX_GEN_* layout below) or null: ideals come from the queue
  int32_t* lite;            // [B][4] {status, q_head, budget, |P|}: what the host polls after a launch, or null
  int32_t done_seq;         // != 0: lite lives in host memory and the host spins on it — the status word carries this
                            // number in bits 17.., written behind a system-scope fence after every other output
  BbxTraceRec* trace;       // [B, trace_stride] or null
  int32_t trace_stride;
  // persistent sessions (bbx_persistent; register/LDS-resident class): asynchronous rollouts queued behind each other do not
  // become one kernel each — the first starts a kernel whose waves keep their environments, and every further call only
  // raises the step total in a device-visible control word the waves look at when they have taken all steps issued so far
  const unsigned long long* ctl;   // control word {bits 0..31: steps issued since the session began, bit 32: stop} or null
  unsigned long long* ctl_stats;   // statistics: steps the closing launches had to take (null: not counted)
  int32_t sess_target;      // != 0: every environment owes sess_target - BbxHdr.sess_done steps (later slices of a session, and
                            // the launch of the HBM-resident class behind them)
  int32_t spill_terms;      // general class, != 0: a merge of more than this many terms hands the environment (untouched: the
                            // step is restartable) to the wide class — one workgroup per environment — launched right behind
  uint32_t slice_ticks;     // persistent kernels: leave after this many ticks of the 100 MHz clock (0: no limit)
  int32_t mbox;             // persistent kernels, != 0: a host MAILBOX session — the control word and the action buffer live in
                            // pinned host memory and are written by the host itself (no launch per step), and every step publishes
                            // its outputs (rewards / dones / rows / observation block, then the status word with the step's
                            // sequence number) to the pinned block the host spins on: bbx_step / bbx_step_obs of small batches
  int32_t wide_hc, wide_fc, wide_rc, wide_sc;   // wide class: LDS capacities (terms) of the polynomial being reduced, the
                                       // reducer-tail window, the reducer table and the accumulator; wide_hc == 0: chosen by the launcher
  // wide class, batches of more than one workgroup per CU: the launch is two kernels.  The first (wide_tail == 1, two
  // workgroups per CU, 128 registers) counts the workgroups that have left in *wide_done; once those still at work would fit
  // one per CU (wide_ncu) each of them stops at its next step boundary (BBX_ST_TIMESLICE, budget kept), and the second
  // (wide_tail == 2: the variant without the register cap, 160 KB of LDS) takes what they owe — a rollout of T steps per
  // environment ends with its slowest environments, and those then run at the speed of a workgroup that has a CU to itself
  int32_t wide_tail, wide_ncu;
  int32_t* wide_done;
};

// Device-side ideal generation (RandomBinomialIdealGenerator, ideals.cpp:156-201): one immutable table per batch, words:
//   [0] n  [1] d  [2] s  [3] flags (1 homogeneous, 2 pure, 4 polynomial distribution = RandomIdealGenerator, ideals.cpp:203-231)
//   [4] #cumulative probabilities (0: degree is always 0)  [5] W  [6..7] exp(-lambda) of the polynomial distribution (double)
//   [BBX_GEN_CP + 2 i]      cumulative probability i of the degree distribution (double, 64 entries, padded with +inf)
//   [BBX_GEN_DEG + 8 i]     per degree i: offset of its monomials (in monomials), their number, the two constants of
//                           libstdc++'s uniform_int_distribution(0, number - 1) — scaling, past — and floor(2^32 / scaling)
//   [BBX_GEN_MONO + W j]    monomial j, packed (with degree slot), in the reference's enumeration order (ideals.cpp:39-64)
#define BBX_GEN_CP 8
#define BBX_GEN_DEG (BBX_GEN_CP + 128)
#define BBX_GEN_MONO (BBX_GEN_DEG + 512)
#define BBX_GEN_MAXDEG 63

// position-keyed commutative hash used for parity traces (same definition in oracle/trace.py)
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint64_t bbx_mix64(uint64_t idx, uint32_t word) {
  uint64_t z = ((idx << 32) | (uint64_t)word) + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// counter-based action hash of the built-in random agent (same in oracle/ffi.py, oracle/*.c*)
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t bbx_agent_hash32(uint32_t seed, uint32_t t) {
  return (uint32_t)(bbx_mix64((uint64_t)seed, t) >> 32);
}
// the row the built-in random agent picks among `rows`: multiply-shift range reduction (one mul_hi on the device)
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
uint32_t bbx_agent_action32(uint32_t seed, uint32_t t, uint32_t rows) {
  return (uint32_t)(((uint64_t)bbx_agent_hash32(seed, t) * rows) >> 32);
}
