// The hand-tuned kernel for the headline class (kernels; bbx_fast.hip holds the launcher):
//   <= 3 variables (8-byte monomials), binomial ideals, Gebauer-Moeller elimination, sorted reducers —
//   i.e. what the reference's C++ LeadMonomialsEnv always runs (buchberger.cpp:377) on 3-20-10-weighted.
//
// One wavefront owns one environment for the whole launch.  Where the state lives:
//   registers  reducer-order arrays (lead monomial, tail monomial, {tc, 1/lc, sugar, basis index}), lane l holds
//              reducers l and l+64: the first-divisor scan is pure VALU + ballot, the chosen reducer is fetched with
//              v_readlane, and the sorted insert is a DPP wave shift (no memory traffic at all in a reduction round)
//   LDS        basis-order arrays (lm, tm, ginfo) and the pair list, which are gathered by per-lane index
//              (pair criteria, observation): 4 KB per environment at compile-time offsets
//   HBM        the record (binomial layout, bbx_common.h) is read at launch start and written back at launch end;
//              observations are written every step
// An environment that outgrows |G| <= 128 / |P| <= 256 leaves a consistent record behind (BBX_ST_SPILL) and is
// continued by the HBM-resident binomial kernel (bbx_binom.h) launched right behind on the same stream.
//
// What makes it fast: everything a wave decides about its environment is wave-uniform and is
// kept provably so for the compiler (scalar branches, the reduction loop on scalar registers); the launch shape that
// is benchmarked and the common |G| < 64 case have their own instantiations without the general cases' code; new
// ideals are drawn right here at reset (gen_binomial in bbx_device.h).
//
// Gebauer-Moeller new pairs without the std::map walk (buchberger.cpp:78-91): the lcms L_i = lcm(LM G_i, LM f) that
// survive are exactly those minimal under divisibility.  They are peeled by increasing degree: all candidates of
// minimal degree are minimal (a proper divisor has strictly smaller degree); each such bucket of equal lcms emits the
// pair of its smallest index unless a member is coprime to f, then every multiple of it is discarded.  Cost is
// proportional to the number of minimal lcms (a handful) instead of |G|^2/64.
#pragma once

typedef Mono<2> M2;
// LDS-class capacities and the per-wave LDS layout: |G| <= 64 NBK, |P| <= 128 NBK.  The first 128 reducers live in
// registers (two banks of 64, FastState); with NBK = 4 reducers 128..255 exist as an ORDER only — their basis indices in
// reducer order, u16 sx[] in LDS — and everything about them is read through the basis-order arrays when it is needed
// (the overflow paths of the reduction and of the sorted insert: cold code, no register state of its own).  NBK = 2
// (4 KB per environment) is what the policy kernels run — their registers and LDS also hold the policy —, NBK = 4 (8.25 KB,
// 16 environments per CU = 132 KB of the CU's 160) everything else: on 3-20-10-weighted a basis passes 128 elements about
// once per 10^5 episodes, and an environment that leaves the class is waited for by every join.
template <int NBK> struct FLay {
  static constexpr int G = 64 * NBK, P = 2 * G, OVF = G > 128 ? G - 128 : 0;
  static constexpr int OFF_LM = 0, OFF_TM = 8 * G, OFF_GI = 16 * G, OFF_PR = 24 * G, OFF_SX = 24 * G + 4 * P, BYTES = OFF_SX + 2 * OVF;
};
constexpr int FNBK_WIDE = 4, FNBK_POL = 2;
constexpr uint32_t FSENT = 0xFFFFFFFFu;               // sentinel monomial word: divides nothing, greater than everything

struct BbxFastParams {
  char* recs; const uint32_t* qwords; const int32_t* qtail; const uint16_t* inv_table;
  const int32_t* actions; double* rewards; uint8_t* dones; int32_t* rows; int32_t* obs; BbxTraceRec* trace;
  uint32_t rec_bytes, hbmG;                           // HBM record stride and its basis capacity (offsets follow from it)
  uint32_t q_env_stride, q_slot_words, q_nslots, q_fixed;
  int32_t B, nsteps, obs_rows, trace_stride, k, nvars, lim_G, lim_P;
  int32_t agent, auto_reset, set_budget, pass, obs_every_step, obs_fill, rewards_mode;
  int32_t* lite;                                      // [B][4] {status, q_head, budget, |P|} for the host, or null
  int32_t done_seq;                                   // see BbxParams::done_seq
  const uint32_t* gen;                                // device-side ideal generator table or null (ideals come from the queue)
  int32_t sort_input;                                 // device-drawn ideals enter in ascending lead-monomial order
  unsigned long long* prof;                           // diagnostic build only: [B][8] cycle sums per phase
  const unsigned long long* ctl; int32_t sess_target; uint32_t slice_ticks;   // persistent sessions: see BbxParams
  double gamma; double* values;                       // VAL instantiation (value(), buchberger.cpp:332-351): discount, [B] returns
  unsigned long long* ctl_stats;                      // statistics word: steps taken by closing launches
  int32_t mbox;                                       // host mailbox session (BbxParams::mbox)
};

__device__ __forceinline__ uint32_t f_readlane(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ M2 f_readlane(const M2& v, int l) { M2 r; r.w[0] = f_readlane(v.w[0], l); r.w[1] = f_readlane(v.w[1], l); return r; }
__device__ __forceinline__ uint2 f_readlane(const uint2& v, int l) { return make_uint2(f_readlane(v.x, l), f_readlane(v.y, l)); }
__device__ __forceinline__ uint32_t f_wave_shr(uint32_t v) {           // lane i <- lane i-1 (lane 0 undefined: fixed by caller)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t f_wave_shl(uint32_t v) {           // lane i <- lane i+1 (lane 63 undefined: fixed by caller)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t f_wave_min(uint32_t x) {
#define FDPPMIN(ctrl, rmask) { uint32_t y_ = (uint32_t)__builtin_amdgcn_update_dpp(-1, (int)x, ctrl, rmask, 0xF, false); x = y_ < x ? y_ : x; }
  FDPPMIN(0x111, 0xF) FDPPMIN(0x112, 0xF) FDPPMIN(0x114, 0xF) FDPPMIN(0x118, 0xF) FDPPMIN(0x142, 0xA) FDPPMIN(0x143, 0xC)
#undef FDPPMIN
  return f_readlane(x, 63);
}
// lane bit of a wave-uniform 64-bit mask as a per-lane predicate (the mask IS an exec-style lane mask: no shifts), and
// the number of set bits below the lane (v_mbcnt pair on the scalar mask)
__device__ __forceinline__ bool f_lane_in(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
__device__ __forceinline__ int f_prefix(uint64_t mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
__device__ __forceinline__ uint64_t f_u64(const M2& m) { return ((uint64_t)m.w[1] << 32) | m.w[0]; }
__device__ __forceinline__ uint64_t f_lowmask(int n) { return n >= 64 ? ~0ull : ((1ull << n) - 1ull); }   // n in [0,64]

// shift-insert `nv` at position p (0..63) of a 64-lane register array; returns the element shifted out of lane 63
__device__ __forceinline__ uint32_t f_insert(uint32_t& arr, uint32_t nv, int p, int lane) {
  uint32_t out = f_readlane(arr, 63);
  uint32_t sh = f_wave_shr(arr);
  arr = lane > p ? sh : (lane == p ? nv : arr);
  return out;
}
// shift the whole array up by one, `carry` enters lane 0; returns the element shifted out of lane 63
__device__ __forceinline__ uint32_t f_shift_in(uint32_t& arr, uint32_t carry, int lane) {
  uint32_t out = f_readlane(arr, 63);
  uint32_t sh = f_wave_shr(arr);
  arr = lane == 0 ? carry : sh;
  return out;
}

// Kernel arguments that only cold code needs (reset, the write-back behind the step loop, error paths) are re-read from
// the kernarg segment where they are used, through a pointer the optimiser cannot see through: otherwise every field is
// loaded once at kernel entry and stays live across the whole step loop, and the scalar register file (the loop keeps
// its polynomial state there) spills to VGPR lanes.  Constant address space + uniform address = s_load.
typedef const __attribute__((address_space(4))) BbxFastParams* FColdParams;
__device__ __forceinline__ FColdParams f_cold_params() {
  FColdParams q = (FColdParams)__builtin_amdgcn_kernarg_segment_ptr();   // the kernels take one by-value struct: offset 0
  asm volatile("" : "+s"(q));
  return q;
}

// the policy kernels' arguments: the step parameters first (f_cold_params() reads them at offset 0), then the policy
struct BbxFastPolicyParams { BbxFastParams f; BbxPolicy pol; };
typedef const __attribute__((address_space(4))) BbxPolicy* FColdPolicy;
__device__ __forceinline__ FColdPolicy f_cold_policy() {
  const __attribute__((address_space(4))) BbxFastPolicyParams* q = (const __attribute__((address_space(4))) BbxFastPolicyParams*)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return &q->pol;
}

// compile-time loop over reducer banks: f(std::integral_constant<int, b>) for b = 0 .. N-1 (every index a constant, so the
// bank arrays below stay registers)
template <class F, int... I> __device__ __forceinline__ void f_banks_(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void f_banks(F&& f) { f_banks_(f, std::make_integer_sequence<int, N>{}); }

constexpr int FRB = 2;             // reducer banks held in registers
struct FastState {                 // reducer-order arrays of the first 128 reducers, lane l <-> reducers l + 64 b of bank b
  M2 slm[FRB], stm[FRB];
  uint2 sin[FRB];                  // .x = tc | (-tc/lc) << 16 ; .y = sugar | basis index << 16
};

// The ordering strategies (Normal, Sugar and the reversed ones: buchberger.cpp:160-199 via select_pair, bbx_device.h) on the
// class's LDS arrays.  Out of line: they are the agents of value() / buchberger() rollouts, not of the timed rollouts, and must
// not weigh on the step loop's registers (until round 4 these agents ran on the LDS-staged class kernel: value('normal') 3.2 M
// values/s against 9.1 M for 'degree').
struct FSelView { const uint32_t* pairs; const Mono<2>* lm; };
__device__ __attribute__((noinline)) int f_select_ordered(const uint32_t* pairs, const Mono<2>* lm, const uint2* gi, int nP, int agent) {
  const FSelView v{pairs, lm};
  return select_pair_inl<2>(v, nP, agent, [gi](int g) { return (int)(gi[g].y >> 16); });
}

// TRACE: per-step parity hashes (tests).  ACCT: count the algorithmic bytes of every step (roofline numerator;
// a property of the workload, so the lean production variant leaves it out and bench.py obtains it from an
// accounting run over a copy of the same batch).
#ifdef BBX_MARK
#define FMARK(slot) asm volatile("; FMARK " #slot)
#else
#define FMARK(slot)
#endif
#define FSTAMP(slot) do { FMARK(slot); if (PROF) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); prof_sum[slot] += t_ - prof_last; prof_last = t_; } } while (0)
// HL: the headline rollout shape as compile-time constants (random-hash agent, 3 variables, k = 2, the observation
// written after every step without fill, auto-reset): no dispatch on launch parameters inside the step loop and fewer
// scalar registers live across it.  The launcher picks it when the parameters say exactly that.
// POL > 0: a policy ROLLOUT (the kernel's arguments are BbxFastPolicyParams): every step starts by writing the
// observation, evaluates the PMLP policy (POL unit blocks of 32, 3 variables and k = 2: 12 columns) on the rows — taken
// straight from the pair list and the monomial arrays in LDS, exactly the values the observation holds — and samples
// its action; per-step outputs go to [nsteps][B] arrays (BbxPolicy).  Waves never wait for each other between steps.
// PERSIST: a kernel of a persistent session (bbx_persistent): the session's first kernel starts with the p.nsteps steps
// issued so far (set_budget), later ones with what the environment still owes of the total (sess_target).  A wave that has
// taken every step issued reads the control word (one relaxed agent-scope atomic load: coherent by the memory model, no
// fences — the environment never changes hands, so nothing else has to become visible to anybody before the kernel ends)
// and carries on if the host has issued more meanwhile.  It leaves when told to stop, after 20 ms without news, or when
// the kernel's time slice is over (steps still owed then: BBX_ST_TIMESLICE) — exits every wave reaches (s_memrealtime
// counts at 100 MHz whatever the shader clock does).  What is owed when it leaves is taken by the session's next kernel.
// VAL: value() rollouts (buchberger.cpp:248-252, 332-351): the discounted return of the steps taken is accumulated in
// double, without fusing multiply and add, and written to values[] when the wave leaves.
// NBK: capacity in units of 64 basis elements (FLay): 2 = registers only, 4 = with the overflow order in LDS.
template <bool TRACE, bool ACCT, bool PROF = false, bool HL = false, int POL = 0, bool PERSIST = false, bool VAL = false, int NBK = (POL > 0 ? FNBK_POL : FNBK_WIDE)>
__device__ __forceinline__ void fast_body(const BbxFastParams& p, char* smem, int ext_action = -1) {
  typedef FLay<NBK> LY;
  constexpr int FG = LY::G, FP = LY::P, FLDS_BYTES = LY::BYTES;
  constexpr int FOFF_LM = LY::OFF_LM, FOFF_TM = LY::OFF_TM, FOFF_GI = LY::OFF_GI, FOFF_PR = LY::OFF_PR;
  constexpr bool OVF = LY::OVF > 0;                        // reducers beyond the register banks (their order in LDS)
  unsigned long long prof_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prof_last = PROF ? __builtin_amdgcn_s_memtime() : 0;
  const int lane = lane_id();
  const int wave_in_block = uni((int)(threadIdx.x / WAVE));
  const int env = blockIdx.x * (blockDim.x / WAVE) + wave_in_block;
  // POL: the prepared weights are the same at every step of the launch: the workgroup stages them in LDS once (behind the
  // per-wave regions), every tile then reads them with ds_read instead of going to L1 / L2
  typedef const __attribute__((address_space(3))) float* PolW;
  PolW pol_w = nullptr;
  if constexpr (POL > 0) {
    const int nfl = (2 * 6 + 2) * 32 * POL + 4;
    __attribute__((address_space(3))) float* wl_ = (__attribute__((address_space(3))) float*)(smem + (blockDim.x / WAVE) * (FLDS_BYTES + 4 * FP));
    const float* wg = f_cold_policy()->wp;
    for (int i = (int)threadIdx.x; i < nfl; i += (int)blockDim.x) wl_[i] = wg[i];
    __syncthreads();                                     // (before any wave may leave)
    pol_w = wl_;
  }
  if (env >= p.B) return;
  char* grec = p.recs + (size_t)env * p.rec_bytes;
  BbxHdr* ghdr = (BbxHdr*)grec;

  int nG = uni(ghdr->nG), nP = uni(ghdr->nP);
  int status = uni(ghdr->status), need_reset = uni(ghdr->need_reset), q_head = uni(ghdr->q_head);
  int t_agent = uni(ghdr->t);
  uint32_t gen_state = ghdr->gen_rng;                  // (a vector register: only the reset touches it)
  uint32_t std_rng = HL ? 0u : ghdr->std_rng;          // (likewise: only the seeded std::default_random_engine selection touches it)
  double vret = 0.0, vdisc = 1.0;                      // VAL
  if (VAL && !p.set_budget) { vret = ghdr->vret; vdisc = ghdr->vdisc; }
  const uint32_t agent_seed = (uint32_t)uni((int)ghdr->agent_seed);
  int budget = uni(ghdr->budget), rollout_pos = uni(ghdr->rollout_pos), done_last = uni(ghdr->done_last);
  if (status == BBX_ST_STARVED || status == BBX_ST_SPILL || status == BBX_ST_TIMESLICE) status = BBX_ST_OK;
  if (p.set_budget) { budget = bbx_st_capacity(status) ? budget + p.nsteps : p.nsteps; rollout_pos = 0; done_last = 0; }   // (bbx_common.h: bbx_st_capacity)
  if (p.sess_target) budget = p.sess_target - uni(ghdr->sess_done);   // later kernels of a persistent session: what is still owed
  const uint32_t t_begin = PERSIST ? (uint32_t)__builtin_amdgcn_s_memrealtime() : 0u;
  int mb_action = -1;                                  // mailbox session of ONE environment: the action that came with the control word
  bool mb_pending = false;                             // mailbox session: a step has been taken whose outputs the host is waiting for
  int pol_t0 = 0;                                      // POL + PERSIST: agent step counter minus session step, fixed for the kernel
  if (POL > 0 && PERSIST) { int vz_; asm volatile("v_mov_b32 %0, 0" : "=v"(vz_)); pol_t0 = vz_ + (t_agent - (p.set_budget ? 0 : uni(ghdr->sess_done))); }
  if (p.pass == 1 && !(status == BBX_ST_OK && (need_reset || (budget > 0 && nP > 0)))) return;

  // HBM record arrays (binomial layout: every array 16-B aligned, capacities hbmG / maxP); the addresses are only
  // formed where the record is read or written (launch start / end), never kept live across the step loop
#define F_HBM_PTRS(PP) \
  const uint32_t HG = (PP)->hbmG; char* grec_ = (PP)->recs + (size_t)env * (PP)->rec_bytes; \
  M2* g_lm = (M2*)(grec_ + 128);            M2* g_tm = (M2*)(grec_ + 128 + 8 * HG); \
  M2* g_slm = (M2*)(grec_ + 128 + 16 * HG); M2* g_stm = (M2*)(grec_ + 128 + 24 * HG); \
  uint2* g_gi = (uint2*)(grec_ + 128 + 40 * HG); uint2* g_si = (uint2*)(grec_ + 128 + 48 * HG); \
  uint32_t* g_pr = (uint32_t*)(grec_ + 128 + 56 * HG);
  // LDS working arrays at compile-time offsets
  char* lbase = smem + wave_in_block * (FLDS_BYTES + (POL > 0 ? 4 * FP : 0));   // (POL: + the logits of up to FP rows)
  M2* lm = (M2*)(lbase + FOFF_LM); M2* tm = (M2*)(lbase + FOFF_TM);
  uint2* gi = (uint2*)(lbase + FOFF_GI); uint32_t* pairs = (uint32_t*)(lbase + FOFF_PR);
  const int limG = p.lim_G < FG ? p.lim_G : FG, limP = p.lim_P < FP ? p.lim_P : FP;

  uint16_t* sx = (uint16_t*)(lbase + LY::OFF_SX);         // OVF: basis indices of reducers 128.. in reducer order
  FastState S;
  auto clear_reducers = [&]() { f_banks<FRB>([&](auto b_) { constexpr int b = decltype(b_)::value; S.slm[b].w[0] = S.slm[b].w[1] = FSENT; }); };
  clear_reducers();
  f_banks<FRB>([&](auto b_) { constexpr int b = decltype(b_)::value; S.stm[b] = m_zero<2>(); S.sin[b] = make_uint2(0, 0); });
  bool staged_in = false;
  // (an environment with nothing to do in this launch — its episode is over and no reset is due — is not staged, whatever its
  // size: handing it to the HBM-resident pass, which has nothing to do for it either, left its outputs unwritten and a host
  // step returned the previous call's row count / reward / done flag for it: found by scripts/fuzz_gym.py)
  const bool idle0 = !PERSIST && !need_reset && nP == 0;
  if (status == BBX_ST_OK && !idle0) {
    if (nG > limG || nP > limP) status = BBX_ST_SPILL;
    else {
      F_HBM_PTRS(&p)
      f_banks<NBK>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        const int i = lane + 64 * b;
        if (i < nG) {
          if constexpr (b < FRB) { S.slm[b] = g_slm[i]; S.stm[b] = g_stm[i]; S.sin[b] = g_si[i]; }
          else sx[i - 64 * FRB] = (uint16_t)(g_si[i].y >> 16);
          lm[i] = g_lm[i]; tm[i] = g_tm[i]; gi[i] = g_gi[i];
        }
      });
      for (int i = lane; i < nP; i += WAVE) pairs[i] = g_pr[i];
      staged_in = true;
      wave_sync();
    }
  }
  // the first 64 pairs also live in a register (lane k <-> pairs[k]; LDS stays the complete copy): the selected pair is
  // a v_readlane, removing it a DPP shift, and the Gebauer-Moeller filter starts its gathers without a pair load
  uint32_t PA = pairs[lane];

  // the random agent's hashes for 64 consecutive steps at a time, one per lane (recomputed every 64 steps)
  uint32_t hv = bbx_agent_hash32(agent_seed, (uint32_t)((t_agent & ~63) + lane));
  // statistics that nothing in the loop branches on live in vector registers (lane-uniform values behind an opaque
  // zero): their updates cost the vector unit, which has slack, instead of scalar instructions and SGPRs, which do not
  int vzero; asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  int adds = vzero, episodes = vzero, zero_red = vzero;
  int trace_pos = rollout_pos;                                        // TRACE only
  long long bytes_total = 0;
  int last_nred = vzero - 1;                           // reward of the last step, kept as its integer reduction count
  int obs_trunc = vzero;                               // an observation had more rows than the caller's block (rows cut)
  const bool tracing = TRACE && p.trace != nullptr;
  const int n = HL ? 3 : p.nvars, kk = HL ? 2 : p.k;
  const int agent = HL ? BBX_AGENT_HASH : p.agent;
  const bool auto_reset = HL ? true : p.auto_reset != 0;
  const bool obs_fill = HL ? false : p.obs_fill != 0;
  // rows [filled_to, obs_rows) of this environment's block already hold the -1 padding: everything after a full fill
  // (obs_fill == 1: unknown contents, so the first write pads the whole block), or — obs_fill == 2, the caller vouches
  // that block and rows[] still hold what the previous call left — everything beyond the row count written then
  int filled_to = (!HL && p.obs_fill == 2 && p.rows) ? uni(p.rows[env]) : 0x7fffffff;
  const bool obs_step = HL ? true : (POL > 0 ? false : (p.obs_every_step && p.obs));
  const int per_row = 2 * kk;
  const int obs_row_bytes = 4 * per_row * n;
  // lane -> (row within a sweep, slot) of the observation matrix, fixed for the launch
  const int o_rl = lane / per_row, o_slot = lane - o_rl * per_row;
  const int rows_per_sweep = per_row <= WAVE ? WAVE / per_row : 0;

  // ---- helpers as lambdas over the state above ----------------------------------------------------------------------
  // The observation of the headline shape (3 variables, k = 2: rows of 4 monomials x 3 exponents = 48 bytes) with
  // everything that depends on the lane decided once per launch: lane -> (row in the sweep, which pair member, lead or
  // tail) and its output address; per trip two pair gathers, two monomial gathers and two 12-byte stores.
  struct __attribute__((aligned(4))) ObsI3 { int32_t a, b, c; };
  const bool obs32 = HL ? true : (n == 3 && kk == 2 && p.obs != nullptr);
  const int o3_row = lane >> 2, o3_hi = (lane >> 1) & 1;
  const char* o3_mono = lbase + ((lane & 1) ? FOFF_TM : FOFF_LM);
  int32_t* const o3_base = obs32 ? p.obs + (size_t)env * p.obs_rows * 12 : nullptr;   // (wave-uniform: the block of this environment)
  typedef int32_t ObsV3 __attribute__((ext_vector_type(3)));
  size_t o3_toff = 0;                                    // POL: the step's slice of a [nsteps][B][rows][12] block
  auto write_obs32 = [&]() {
    const int rows = nP < p.obs_rows ? nP : p.obs_rows;
    // The rows go out through a buffer descriptor whose size is exactly the live part of the block: the hardware drops the
    // stores of lanes beyond it, so no lane is ever masked — the exec-mask bookkeeping of predicated gathers and stores
    // was ~16 scalar instructions per 32 rows, on the unit that binds this kernel.  The gathers run for all lanes: a pair
    // index beyond |P| stays inside the pair array (|P| <= FP = its capacity, and a trip covers rows r0 .. r0 + 31 with
    // r0 <= FP - 32), what it reads is a stale pair, and the basis index taken from it is clamped to the arrays' FG entries.
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(o3_base + o3_toff), 0, rows * 48, 0x00020000);
    for (int r0 = 0; r0 < rows; r0 += 32) {
      const int ra = r0 + o3_row, rb = ra + 16;
      const uint32_t pa = pairs[ra], pb = pairs[rb];
      const uint32_t ga = (o3_hi ? pa >> 16 : pa) & (uint32_t)(FG - 1), gb = (o3_hi ? pb >> 16 : pb) & (uint32_t)(FG - 1);
      const M2 ma = *(const M2*)(o3_mono + ga * 8), mb = *(const M2*)(o3_mono + gb * 8);
      const ObsV3 va = {(int32_t)(ma.w[0] & 0xffffu), (int32_t)(ma.w[0] >> 16), (int32_t)(ma.w[1] & 0xffffu)};
      const ObsV3 vb = {(int32_t)(mb.w[0] & 0xffffu), (int32_t)(mb.w[0] >> 16), (int32_t)(mb.w[1] & 0xffffu)};
      __builtin_amdgcn_raw_buffer_store_b96(va, rs, r0 * 48 + lane * 12, 0, 0);
      __builtin_amdgcn_raw_buffer_store_b96(vb, rs, r0 * 48 + lane * 12 + 768, 0, 0);
    }
    if (obs_fill) {
      int32_t* out = p.obs + (size_t)env * p.obs_rows * 12;
      const int hi = filled_to < p.obs_rows ? filled_to : p.obs_rows;
      for (int idx = rows * 12 + lane; idx < hi * 12; idx += WAVE) out[idx] = -1;
      filled_to = rows;
    }
  };
  auto write_obs = [&](bool write, bool want_hash) -> uint64_t {
    const int cols = per_row * n;
    int32_t* out = (write && p.obs) ? p.obs + (size_t)env * p.obs_rows * cols : nullptr;
    const int rows = out ? (nP < p.obs_rows ? nP : p.obs_rows) : nP;
    uint64_t h = 0;
    if (rows_per_sweep > 0) {
      const int half = o_slot >= kk ? 1 : 0, t = o_slot - half * kk;
      const bool lane_on = o_rl < rows_per_sweep;
      auto emit = [&](int r, const M2& mm) {
        const int base = (r * per_row + o_slot) * n;
        const uint32_t e0 = mm.w[0] & 0xffffu, e1 = mm.w[0] >> 16, e2 = mm.w[1] & 0xffffu;
        if (out) {
          if (n == 3) { int3 v3 = make_int3((int)e0, (int)e1, (int)e2); *(int3*)(out + base) = v3; }
          else { out[base] = (int)e0; if (n > 1) out[base + 1] = (int)e1; }
        }
        if (TRACE && want_hash) {
          h += bbx_mix64((uint64_t)base, e0);
          if (n > 1) h += bbx_mix64((uint64_t)(base + 1), e1);
          if (n > 2) h += bbx_mix64((uint64_t)(base + 2), e2);
        }
      };
      // two sweeps per trip: both pair loads, then both monomial loads, are in flight together
      for (int r0 = 0; r0 < rows; r0 += 2 * rows_per_sweep) {
        const int ra = r0 + o_rl, rb = ra + rows_per_sweep;
        const bool oa = lane_on && ra < rows, ob = lane_on && rb < rows;
        const uint32_t pa = oa ? pairs[ra] : 0u, pb = ob ? pairs[rb] : 0u;
        const int ga = half ? (int)(pa >> 16) : (int)(pa & 0xffffu), gb = half ? (int)(pb >> 16) : (int)(pb & 0xffffu);
        M2 ma = m_zero<2>(), mb = m_zero<2>();
        if (t == 0) { ma = lm[ga]; mb = lm[gb]; } else if (t == 1) { ma = tm[ga]; mb = tm[gb]; }   // tm is zero when G has no tail
        if (oa) emit(ra, ma);
        if (ob) emit(rb, mb);
      }
    } else {
      for (int it = lane; it < rows * per_row; it += WAVE) {
        const int r = it / per_row, slot = it - r * per_row;
        const int half = slot >= kk ? 1 : 0, t = slot - half * kk;
        const uint32_t pr = pairs[r];
        const int g = half ? (int)(pr >> 16) : (int)(pr & 0xffffu);
        M2 mm = m_zero<2>();
        if (t == 0) mm = lm[g]; else if (t == 1) mm = tm[g];
        for (int v = 0; v < n; v++) {
          uint32_t x = m_exp(mm, v);
          if (out) out[it * n + v] = (int)x;
          if (TRACE && want_hash) h += bbx_mix64((uint64_t)(it * n + v), x);
        }
      }
    }
    if (out && obs_fill) {
      const int hi = filled_to < p.obs_rows ? filled_to : p.obs_rows;
      for (int idx = rows * cols + lane; idx < hi * cols; idx += WAVE) out[idx] = -1;
      filled_to = rows;
    }
    return (TRACE && want_hash) ? wave_sum64(h) : 0;
  };

  // append the binomial (t0, t1) to the basis: G-order arrays, Gebauer-Moeller update, sorted reducer insert
  // (buchberger.cpp:52-99 + 321-326).  The caller has checked the capacities.
  // `skip` >= 0: the pair at that index has just been selected and is removed by the same compaction pass
  // (P.erase(remove(action)), buchberger.cpp:319) instead of a separate shift of the list
  // `banks` (a std::integral_constant<int, BK>): the caller guarantees |G| < 64 BK.  BK = 1 (the common case: resets of
  // ideals with at most 64 generators, bases below 64 elements): everything about the other banks of the reducer / basis
  // registers compiles away; BK = 2, 4: banks beyond the first are guarded by wave-uniform tests on |G|; BK = 4 (|G| >= 128:
  // cold) additionally maintains the overflow order sx[].
  auto add_poly = [&](const BTerm<2>& t0, const BTerm<2>& t1, int sugar, int skip, auto banks) {
    constexpr int BK = decltype(banks)::value;
    const int g = nG;                                     // == m of update()
    // 1/LC comes from a table in HBM/L2: issue the load now, consume it at the very end (the pair update below
    // does not need it), so its latency overlaps the Gebauer-Moeller work
    const uint32_t inv_raw = t0.c == 1 ? 1u : (uint32_t)p.inv_table[t0.c];
    const M2 f = t0.m;
    const M2 tail = t1.c ? t1.m : m_zero<2>();
    if (lane == 0) { lm[g] = f; tm[g] = tail; }
    // (70-76) drop old pairs (i,j): LM f | lcm_ij and lcm_ij != lcm_if and lcm_ij != lcm_jf.  Only exponents matter,
    // so the lcms are raw v_pk_max words with the degree slot masked out of the comparisons.
    // The gathers of the first 64 pairs are issued here and consumed behind the register-only peel below, which
    // hides their two dependent LDS round trips.
    const int nP_old = nP;
    const uint32_t pr_first = lane < nP_old ? PA : 0u;
    const M2 li_first = lm[pr_first & 0xffffu], lj_first = lm[pr_first >> 16];
    // (78-91) new pairs (i, g): minimal lcms by degree peeling
    uint64_t emit[BK];
    {
      M2 L[BK]; uint64_t valid[BK], cp[BK], cand[BK]; uint32_t d[BK];
      f_banks<BK>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        const M2 l = lm[lane + 64 * b];                    // basis order; lanes >= g hold garbage (masked by valid)
        if constexpr (b == 0) valid[0] = f_lowmask((BK == 1 || g < 64) ? g : 64);
        else valid[b] = g > 64 * b ? f_lowmask(g - 64 * b) : 0ull;
        L[b] = m_lcm(l, f);
        if constexpr (b == 0) cp[0] = ballot64(m_coprime(l, f)) & valid[0];
        else cp[b] = valid[b] ? (ballot64(m_coprime(l, f)) & valid[b]) : 0ull;
        d[b] = L[b].w[1] >> 16; cand[b] = valid[b]; emit[b] = 0;
      });
      auto any = [&](const uint64_t (&m)[BK]) { uint64_t o = 0; f_banks<BK>([&](auto b_) { o |= m[decltype(b_)::value]; }); return o; };
      while (any(cand)) {
        uint32_t dm = f_lane_in(cand[0]) ? d[0] : 0xFFFFFFFFu;
        f_banks<BK>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if constexpr (b > 0) if (cand[b]) { const uint32_t t = f_lane_in(cand[b]) ? d[b] : 0xFFFFFFFFu; dm = t < dm ? t : dm; }
        });
        const uint32_t dmin = f_wave_min(dm);
        uint64_t surv[BK];
        f_banks<BK>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if constexpr (b == 0) surv[0] = ballot64(f_lane_in(cand[0]) && d[0] == dmin);
          else surv[b] = cand[b] ? ballot64(f_lane_in(cand[b]) && d[b] == dmin) : 0ull;
        });
        while (any(surv)) {
          M2 Ls; int s = 0, sb = 0; bool got = false;      // the first survivor: lane s of bank sb
          f_banks<BK>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            if (!got && (b == BK - 1 || surv[b])) { s = __builtin_ctzll(surv[b]); Ls = f_readlane(L[b], s); sb = b; got = true; }
          });
          const uint64_t ls = f_u64(Ls);
          uint64_t eq[BK], dv[BK], bad = 0;
          f_banks<BK>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            if constexpr (b == 0) { eq[0] = ballot64(f_u64(L[0]) == ls); dv[0] = ballot64(m_divides(Ls, L[0])); bad |= eq[0] & valid[0] & cp[0]; }
            else {
              eq[b] = 0; dv[b] = 0;
              if (valid[b]) { eq[b] = ballot64(f_u64(L[b]) == ls) & valid[b]; dv[b] = ballot64(m_divides(Ls, L[b])); }
              bad |= eq[b] & cp[b];
            }
          });
          if (bad == 0) f_banks<BK>([&](auto b_) { constexpr int b = decltype(b_)::value; if (BK == 1 || sb == b) emit[b] |= 1ull << s; });
          f_banks<BK>([&](auto b_) { constexpr int b = decltype(b_)::value; surv[b] &= ~eq[b]; cand[b] &= ~dv[b]; });
        }
      }
    }
    int w = 0;
    for (int base = 0; base < nP_old; base += WAVE) {
      const int k = base + lane;
      const bool valid = k < nP_old;
      uint32_t pr; M2 li, lj;
      if (base == 0) { pr = pr_first; li = li_first; lj = lj_first; }
      else { pr = valid ? pairs[k] : 0u; li = lm[pr & 0xffffu]; lj = lm[pr >> 16]; }
      const uint32_t a0 = pk_max(li.w[0], lj.w[0]), a1 = pk_max(li.w[1], lj.w[1]);
      const uint32_t b0 = pk_max(li.w[0], f.w[0]), b1 = pk_max(li.w[1], f.w[1]);
      const uint32_t c0 = pk_max(lj.w[0], f.w[0]), c1 = pk_max(lj.w[1], f.w[1]);
      const bool fdiv = (pk_subsat(f.w[0], a0) | (pk_subsat(f.w[1], a1) & 0xffffu)) == 0;
      const bool eqi = ((a0 ^ b0) | ((a1 ^ b1) & 0xffffu)) == 0;
      const bool eqj = ((a0 ^ c0) | ((a1 ^ c1) & 0xffffu)) == 0;
      const bool keep = valid && k != skip && !(fdiv && !eqi && !eqj);
      const uint64_t mask = ballot64(keep);
      if (keep) pairs[w + f_prefix(mask)] = pr;
      w += __popcll(mask);
    }
    nP = w;
    // (92) ascending i, appended behind the surviving old pairs (98)
    f_banks<BK>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      if (b == 0 || emit[b]) {
        if (f_lane_in(emit[b])) pairs[nP + f_prefix(emit[b])] = (uint32_t)(lane + 64 * b) | ((uint32_t)g << 16);
        nP += __popcll(emit[b]);
      }
    });
    // sorted reducer insert: std::upper_bound by lead monomial (buchberger.cpp:323-324); sentinels compare greater
    {
      constexpr int RB = BK < FRB ? BK : FRB;              // register banks this call can reach
      int pos = __popcll(ballot64(!m_gt(S.slm[0], f)));
      f_banks<RB>([&](auto b_) {
        constexpr int b = decltype(b_)::value;
        if constexpr (b > 0) if (g >= 64 * b) pos += __popcll(ballot64(!m_gt(S.slm[b], f)));
      });
      const int novf = (OVF && BK > FRB && g > 64 * FRB) ? g - 64 * FRB : 0;   // reducers beyond the registers (cold)
      if (OVF && BK > FRB) for (int k0 = 0; k0 < novf; k0 += WAVE) {
        const int k = k0 + lane;
        const bool le = k < novf && !m_gt(lm[sx[k < novf ? k : 0]], f);
        pos += __popcll(ballot64(le));
      }
      const uint32_t inv = inv_raw;
      if (lane == 0) gi[g] = make_uint2(t0.c | (t1.c << 16), inv | ((uint32_t)sugar << 16));
      const uint2 ns = make_uint2(t1.c | (negmod(mulmod(t1.c, inv)) << 16), (uint32_t)sugar | ((uint32_t)g << 16));   // .x = tc | (-tc / lc) << 16
      // the overflow order takes the new element (pos beyond the registers) or the reducer that falls out of the last bank
      int ovq = -1; uint32_t ovg = 0;
      if (OVF && BK > FRB && g >= 64 * FRB) {
        if (pos >= 64 * FRB) { ovq = pos - 64 * FRB; ovg = (uint32_t)g; }
        else { ovq = 0; ovg = f_readlane(S.sin[FRB - 1].y, 63) >> 16; }
      }
      // bank pos / 64 takes the new element at lane pos % 64; what falls out of a bank's lane 63 enters the next one's lane 0
      if (!(OVF && BK > FRB) || pos < 64 * FRB) {
        const int pb = RB == 1 ? 0 : pos >> 6, q = RB == 1 ? pos : pos & 63;
        uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
        f_banks<RB>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          if (RB == 1 || pb == b) {
            c0 = f_insert(S.slm[b].w[0], f.w[0], q, lane); c1 = f_insert(S.slm[b].w[1], f.w[1], q, lane);
            c2 = f_insert(S.stm[b].w[0], tail.w[0], q, lane); c3 = f_insert(S.stm[b].w[1], tail.w[1], q, lane);
            c4 = f_insert(S.sin[b].x, ns.x, q, lane); c5 = f_insert(S.sin[b].y, ns.y, q, lane);
          } else if (b > 0 && pb < b && g >= 64 * b) {
            c0 = f_shift_in(S.slm[b].w[0], c0, lane); c1 = f_shift_in(S.slm[b].w[1], c1, lane);
            c2 = f_shift_in(S.stm[b].w[0], c2, lane); c3 = f_shift_in(S.stm[b].w[1], c3, lane);
            c4 = f_shift_in(S.sin[b].x, c4, lane); c5 = f_shift_in(S.sin[b].y, c5, lane);
          }
        });
      }
      if (OVF && BK > FRB && ovq >= 0) {                   // sx[ovq ..] moves up by one, ovg enters at ovq
        for (int k0 = ((novf - 1 - ovq) / WAVE) * WAVE; k0 >= 0; k0 -= WAVE) {   // highest chunk first: no element is overwritten before it moved
          const int k = ovq + k0 + lane;
          uint16_t v = 0;
          if (k < novf) v = sx[k];
          wave_sync();
          if (k < novf) sx[k + 1] = v;
          wave_sync();
        }
        if (lane == 0) sx[ovq] = (uint16_t)ovg;
      }
    }
    nG = g + 1;
    wave_sync();
    PA = pairs[lane];                                      // (consumed by the next step: the read has a whole phase to land)
  };

  for (;;) {
    // |P| is wave-uniform by construction, but the compiler's uniformity analysis loses that across the loop; pinned
    // here, every branch of the step is a scalar branch (without it the whole body runs under exec masks, with
    // per-lane copies of all state at each join: +40 % instructions)
    nP = uni(nP);
    if (status != BBX_ST_OK) break;
    if (need_reset) {                                      // BuchbergerEnv::reset from the next queued ideal(s)
      bool ok = true;
      const FColdParams cq = f_cold_params();
      const uint32_t* gtab = cq->gen;
      if (gtab) {                                          // draw the ideal here (see gen_binomial): no queue, no host
        const int npoly = (int)ldc(gtab + 2), ncp = (int)ldc(gtab + 4);
        const uint32_t gflags = ldc(gtab + 3);
        const GenLanes GL = gen_lanes(gtab);
        uint32_t x = (uint32_t)uni((int)gen_state);
        for (;;) {
          const uint32_t x_start = x;
          nG = 0; nP = 0;
          clear_reducers();
          // sort_input: all generators are drawn first (lane f keeps generator f), then enter in sorted order
          const bool sorted = cq->sort_input != 0;
          M2 tabL = m_zero<2>(), tabT = m_zero<2>(); uint32_t tabC = 0; int rank = 0;
          if (sorted) {
            for (int fidx = 0; fidx < npoly && ok; fidx++) {
              M2 lead, tail; uint32_t c;
              if (!gen_binomial<2>(x, gtab, GL, gflags, ncp, lead, tail, c)) { status = BBX_ST_GEN_FAIL; ok = false; break; }
              if (lane == fidx) { tabL = lead; tabT = tail; tabC = c; }
            }
            rank = gen_sorted_rank<2>(tabL, npoly);
          }
          for (int fidx = 0; fidx < npoly && ok; fidx++) {
            if (nG + 1 > limG || nP + nG > limP) { status = BBX_ST_SPILL; ok = false; x = x_start; break; }   // redone from the same draw
            BTerm<2> t0, t1;
            t0.c = 1;
            if (sorted) {
              const int src = __builtin_ctzll(ballot64(lane < npoly && rank == fidx));
              t0.m = f_readlane(tabL, src); t1.m = f_readlane(tabT, src); t1.c = f_readlane(tabC, src);
            } else if (!gen_binomial<2>(x, gtab, GL, gflags, ncp, t0.m, t1.m, t1.c)) { status = BBX_ST_GEN_FAIL; ok = false; break; }
            if (__builtin_expect(npoly <= 64, 1)) add_poly(t0, t1, (int)m_deg(t0.m), -1, std::integral_constant<int, 1>{});
            else add_poly(t0, t1, (int)m_deg(t0.m), -1, std::integral_constant<int, NBK>{});
          }
          if (!ok || nP != 0) break;                       // buchberger.cpp:313-314: redraw while the pair set is empty
        }
        gen_state = x;
      } else {
      const uint32_t q_slot_words = cq->q_slot_words, q_fixed = cq->q_fixed;
      for (;;) {
        const uint32_t* slot;
        if (q_fixed) slot = cq->qwords;
        else {
          const int tail = uni(cq->qtail[env]);          // (a plain load would count as divergent and drag nG / nP into VGPRs)
          if (q_head >= tail) { status = BBX_ST_STARVED; ok = false; break; }
          slot = cq->qwords + (size_t)env * cq->q_env_stride + (size_t)(q_head % (int)cq->q_nslots) * q_slot_words;
        }
        nG = 0; nP = 0;
        clear_reducers();
        // the whole ideal (<= 128 words for up to 15 binomials) comes in with two coalesced loads, one word per
        // lane; fields are then picked with v_readlane instead of a chain of dependent scalar-address loads
        const bool small_slot = q_slot_words <= 128;
        uint32_t qA = 0, qB = 0;
        if (small_slot) {
          if (lane < (int)q_slot_words) qA = slot[lane];
          if (lane + 64 < (int)q_slot_words) qB = slot[lane + 64];
        }
        auto qword = [&](int j) -> uint32_t {
          if (!small_slot) return (uint32_t)uni((int)slot[j]);
          return j < 64 ? f_readlane(qA, j) : f_readlane(qB, j - 64);
        };
        const int npoly = (int)qword(0);
        int at = 1;
        for (int fidx = 0; fidx < npoly; fidx++) {
          const int nt = (int)qword(at), sugar = (int)qword(at + 1);
          if (nG + 1 > limG || nP + nG > limP) { status = BBX_ST_SPILL; ok = false; break; }
          BTerm<2> t0, t1;
          t0.c = qword(at + 2); t0.m.w[0] = qword(at + 3); t0.m.w[1] = qword(at + 4);
          t1.c = 0; t1.m = m_zero<2>();
          if (nt == 2) { t1.c = qword(at + 5); t1.m.w[0] = qword(at + 6); t1.m.w[1] = qword(at + 7); }
          add_poly(t0, t1, sugar, -1, std::integral_constant<int, NBK>{});
          at += 2 + nt * 3;
        }
        if (!ok) break;
        if (!q_fixed) q_head++;
        if (nP != 0 || q_fixed) break;                   // buchberger.cpp:313-314: redraw while the pair set is empty
      }
      }
      if (!ok) { if (status != BBX_ST_STARVED) { nG = 0; nP = 0; } break; }
      need_reset = 0;
      if constexpr (POL > 0 && PERSIST) {
        // per-step policy calls served by a session (bbx_policy_step_device): the block and the row count the caller finds when
        // the session ends are those of the NEW episode, as after a launch per step — the step that ended the episode wrote the
        // state it left (no rows), and within the session nobody reads the block (found by scripts/fuzz_sessions.py)
        const FColdPolicy polr = f_cold_policy();
        if (polr->post_obs) {
          if (p.obs) { write_obs32(); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
          if (lane == 0 && polr->rows_t) polr->rows_t[env] = nP;
        }
      }
    }
    FSTAMP(0);                                             // 0: loop top / reset
    if constexpr (PERSIST && POL == 0 && !HL) {
      // A host mailbox session: the host spins on this environment's status word for the step's sequence number.  The step's
      // outputs reach host memory first — here, behind the reset of an environment whose episode the step ended (auto-reset:
      // the observation and the row count the call returns are the new episode's, as in a launch per step).
      if (mb_pending) {
        mb_pending = false;
        const FColdParams cm = f_cold_params();
        if (done_last && auto_reset && cm->obs) { if (obs32) write_obs32(); else write_obs(true, false); obs_trunc |= nP > cm->obs_rows ? 1 : 0; }
        __threadfence_system();
        if (lane == 0) {
          const BbxHdr* hh = (const BbxHdr*)(cm->recs + (size_t)env * cm->rec_bytes);
          const int taken = (cm->set_budget ? 0 : hh->sess_done) + t_agent - hh->t;      // steps of the session's total taken so far
          if (cm->rewards) cm->rewards[env] = cm->rewards_mode == BBX_REW_ADDITIONS ? (-1.0 - (double)last_nred) : -1.0;
          if (cm->dones) cm->dones[env] = (uint8_t)done_last;
          if (cm->rows) cm->rows[env] = nP;
          int32_t* lw = cm->lite + 4 * (size_t)env;
          lw[1] = q_head; lw[2] = budget; lw[3] = nP;
          __threadfence_system();
          __hip_atomic_store(lw, BBX_ST_OK | (obs_trunc ? BBX_LITE_OBS_TRUNC : 0) | (((taken % 16000) + 1) << 17), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
    }
    if (budget <= 0) {
      if constexpr (PERSIST) {
        // every step issued so far is taken: have more been issued meanwhile?  (Nothing about the session is kept live
        // across the step loop: the steps taken are the agent's step counter minus what the record's header — untouched
        // until this wave leaves — says it was when the session began.)
        const FColdParams cq = f_cold_params();
        const unsigned long long* ctl = cq->ctl;
        const BbxHdr* hh = (const BbxHdr*)(cq->recs + (size_t)env * cq->rec_bytes);
        const int taken = (cq->set_budget ? 0 : uni(hh->sess_done)) + t_agent - uni(hh->t);   // of the session's total
        const uint32_t t0 = (uint32_t)__builtin_amdgcn_s_memrealtime();
        bool more = false;
        for (;;) {
          // (a mailbox session's word lives in host memory and is followed by the step's action: acquire, system scope)
          const unsigned long long w = cq->mbox ? __hip_atomic_load(ctl, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM)
                                                : __hip_atomic_load(ctl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int tgt = uni((int)(uint32_t)w), stop = uni((int)(uint32_t)(w >> 32));
          // (a mailbox session of one environment carries the step's action in bits 33.. of the word, + 1: one read over the bus
          // instead of two; valid for the LAST step issued, i.e. when exactly one step is owed)
          if (tgt > taken) { budget = tgt - taken; more = true; mb_action = (cq->mbox && budget == 1) ? uni((int)(uint32_t)(w >> 33)) - 1 : -1; break; }
          if (stop & 1) break;
          const uint32_t now = (uint32_t)__builtin_amdgcn_s_memrealtime();
          if (now - t0 > 2000000u) break;                                // 20 ms without news
          if (cq->slice_ticks && now - t_begin > cq->slice_ticks) break; // the slice is over anyway
          if (cq->mbox) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(32);   // (a host waits on the other side: short naps)
        }
        if (!more) break;
      } else break;
    }
    if (nP == 0) break;
    if (nG + 1 > limG || nP - 1 + nG > limP) { status = BBX_ST_SPILL; break; }   // before anything is modified

    // ---- choose the pair -----------------------------------------------------------------------------------------
    int action;
    int pol_tt = 0;
    if constexpr (POL > 0) {
      const FColdPolicy pol = f_cold_policy();
      // step of the rollout (the budget was set to nsteps; closing launches of a session: what is owed of its total), or of
      // the session (PERSIST: the agent's step counter against what it was when the session began)
      pol_tt = PERSIST ? uni(t_agent - pol_t0) : uni((p.sess_target ? p.sess_target : p.nsteps) - budget);
      const size_t tb = (size_t)pol_tt * (size_t)p.B + (size_t)env;
      const float uu = pol->u[tb];                           // (requested before the observation goes out)
      const bool pre_obs = pol->post_obs == 0;
      if (p.obs && pre_obs) { o3_toff = (size_t)pol_tt * (size_t)pol->obs_tstride; write_obs32(); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
      if (lane == 0 && pol->rows_t && pre_obs) pol->rows_t[tb] = nP;
      const PolW wp = pol_w;
      const int plr = lane & 31, plk = lane >> 5;
      float* lg = (float*)(lbase + FLDS_BYTES);
      const float b2 = wp[(2 * 6 + 2) * 32 * POL];
      const int pn = (p.obs && nP > p.obs_rows) ? p.obs_rows : nP;   // the rows the policy sees = the rows of the block
      for (int r0 = 0; r0 < pn; r0 += 32) {
        const int r = r0 + plr;
        const uint32_t prw = r < pn ? pairs[r] : 0u;
        const M2 a0 = lm[prw & 0xffffu], a1 = tm[prw & 0xffffu], c0 = lm[prw >> 16], c1 = tm[prw >> 16];
        // row = [lm_i | tm_i | lm_j | tm_j] x (e0, e1, e2); my k-step operands are columns 2 s + (lane >> 5)
        const uint32_t ev[6] = {plk ? a0.w[0] >> 16 : a0.w[0] & 0xffffu,  plk ? a1.w[0] & 0xffffu : a0.w[1] & 0xffffu,
                                plk ? a1.w[1] & 0xffffu : a1.w[0] >> 16,  plk ? c0.w[0] >> 16 : c0.w[0] & 0xffffu,
                                plk ? c1.w[0] & 0xffffu : c0.w[1] & 0xffffu, plk ? c1.w[1] & 0xffffu : c1.w[0] >> 16};
        float xa[6];
#pragma unroll
        for (int s2 = 0; s2 < 6; s2++) xa[s2] = (float)ev[s2];
        const float logit = pmlp_tile<POL, 6, 1, PolW>(xa, wp, plr, plk);
        if (plk == 0 && r < pn) lg[r] = logit + b2;
      }
      wave_sync();
      action = pmlp_sample(lg, pn, env, uu, pol->actions + (size_t)pol_tt * (size_t)pol->stride_out, pol->logprobs + (size_t)pol_tt * (size_t)pol->stride_out);
    } else
    if (agent == BBX_AGENT_HASH) action = (int)(((uint64_t)f_readlane(hv, t_agent & 63) * (uint32_t)nP) >> 32);   // bbx_agent_action32
    else if (agent == BBX_AGENT_EXTERNAL) {
      const FColdParams ca = f_cold_params();
      if (PERSIST && ca->mbox) {                           // (host memory, rewritten per step)
        if (mb_action >= 0) { action = mb_action; mb_action = -1; }
        else action = uni(__hip_atomic_load(ca->actions + env, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
      }
      else action = ext_action >= 0 ? ext_action : uni(ca->actions[env]);
    }
    else if (agent == BBX_AGENT_FIRST) action = 0;
    else if (!HL && agent == BBX_AGENT_STDRANDOM) {        // choice(P.begin(), P.end(), rng) of the seeded engine (buchberger.cpp:200-203, 244)
      uint32_t x = (uint32_t)uni((int)std_rng);
      action = std_choice(x, nP);
      std_rng = x;
    }
    else if (!HL && agent >= BBX_AGENT_NORMAL && agent <= BBX_AGENT_SPICE) action = uni(f_select_ordered(pairs, lm, gi, nP, agent));
    else {                                                 // degree: first row of minimal deg lcm (buchberger.cpp:171-176)
      uint32_t best = 0xFFFFFFFFu;
      for (int r = lane; r < nP; r += WAVE) {
        const uint32_t pr = pairs[r];
        const uint32_t key = (m_deg(m_lcm(lm[pr & 0xffffu], lm[pr >> 16])) << 16) | (uint32_t)r;   // deg < 2^16, r < 2^16
        best = key < best ? key : best;
      }
      action = (int)(f_wave_min(best) & 0xffffu);
    }
    // (the hash agent's action is floor(hash * |P| / 2^32): in range by construction)
    if (!(HL || agent == BBX_AGENT_HASH) && (action < 0 || action >= nP)) { status = BBX_ST_BAD_ACTION; break; }
    uint32_t pr = f_readlane(PA, action & 63);             // the first 64 pairs are in the register copy
    if (action >= 64) pr = (uint32_t)uni((int)pairs[action]);
    const int gi_ = pr & 0xffffu, gj_ = pr >> 16;            // the pair leaves P below, fused with the update's compaction
    FSTAMP(1);                                             // 1: agent + pair removal
    // ---- S-polynomial (buchberger.cpp:18-21): the lead terms cancel, the scaled tails remain ------------------------
    BTerm<2> h0, h1;
    int hsug, bytes;
    {
      const M2 lmi = lm[gi_], lmj = lm[gj_];
      const uint2 ii = gi[gi_], ij = gi[gj_];
      const M2 gamma = m_lcm(lmi, lmj);
      const M2 si = m_div(gamma, lmi), sj = m_div(gamma, lmj);
      BTerm<2> a, b;
      const uint32_t tci = ii.x >> 16, tcj = ij.x >> 16;
      a.c = tci ? mulmod(tci, ii.y & 0xffffu) : 0u;           a.m = m_mul(tm[gi_], si);
      b.c = tcj ? negmod(mulmod(tcj, ij.y & 0xffffu)) : 0u;   b.m = m_mul(tm[gj_], sj);
      const int sgi = (int)(ii.y >> 16) + (int)m_deg(si), sgj = (int)(ij.y >> 16) + (int)m_deg(sj);
      hsug = uni(sgi > sgj ? sgi : sgj);
      if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; break; }
      merge2<2>(a, b, h0, h1);
      bytes = ACCT ? 12 * ((tci ? 2 : 1) + (tcj ? 2 : 1) + (h0.c ? 1 : 0) + (h1.c ? 1 : 0)) : 0;
    }

    FSTAMP(2);                                             // 2: S-polynomial
    // ---- reduce (buchberger.cpp:24-49), entirely in registers -----------------------------------------------------------
    // (Measured alternatives, DESIGN.md section 4: the same loop in select form on the vector unit — v_cndmask instead of
    // branches — is 12-15 % slower, on the scalar unit 25 % slower: the branches skip work, and 32-bit multiplies of the
    // modular arithmetic are quarter-rate on the vector unit.)
    // h and r are wave-uniform, so they are carried as scalars (SGPRs): the loop's control flow is then scalar
    // branches (no exec masking, no per-lane copies at the joins) and its arithmetic runs on the scalar unit; only
    // the divisor scan over the reducer registers and the v_readlane of the chosen reducer are vector instructions.
    // Monomial quotient / product are plain word subtracts / adds here: no field borrows (the divisor divides) and
    // no field carries while degrees stay below 2^16, which the sugar bound checked below guarantees.
    uint32_t h0c = (uint32_t)uni((int)h0.c), h0a = (uint32_t)uni((int)h0.m.w[0]), h0b = (uint32_t)uni((int)h0.m.w[1]);
    uint32_t h1c = (uint32_t)uni((int)h1.c), h1a = (uint32_t)uni((int)h1.m.w[0]), h1b = (uint32_t)uni((int)h1.m.w[1]);
    uint32_t r0c = 0, r0a = 0, r0b = 0, r1c = 0, r1a = 0, r1b = 0;
    int nred = 0, rsug = 0;
    // One round of the reduction loop in C++ — the specification of the assembly below, the accounting variants' loop, and
    // what serves the rounds whose divisor may lie beyond the reducer registers.  `regs`: scan the register banks (the
    // assembly has done that already when it hands a round over).
    const int ngu = uni(nG);                               // (|G| is wave-uniform; pinned, so that what depends on it stays scalar)
    auto round_cpp = [&](auto regs) {
      const int hn = h1c ? 2 : 1;
      M2 hm; hm.w[0] = h0a; hm.w[1] = h0b;
      int found = -1;
      uint32_t nb0 = 0, nb1 = 0, sxw = 0;
      int fs = 0;
      // Every lane prepares what the round needs from ITS reducer, should it be the divisor — the quotient LT h / LT f,
      // the monomial of the new term tail(f) * quotient, the sugar candidate: vector work the SIMDs have room for — and
      // the chosen lane's results travel by v_readlane: 4 words instead of 6 plus their arithmetic on the scalar unit,
      // which is the unit this kernel is bound by.  (Garbage in lanes that do not divide: unused.)
      if constexpr (decltype(regs)::value) f_banks<FRB>([&](auto b_) {    // banks in reducer order; the first divisor wins
        constexpr int b = decltype(b_)::value;
        if (found < 0 && (b == 0 || ngu > 64 * b)) {
          const uint64_t mb = ballot64(m_divides(S.slm[b], hm));        // sentinels never divide
          if (mb) {
            const int l = __builtin_ctzll(mb);
            found = 64 * b + l;
            const uint32_t q1 = h0b - S.slm[b].w[1];
            nb0 = f_readlane(S.stm[b].w[0] + (h0a - S.slm[b].w[0]), l); nb1 = f_readlane(S.stm[b].w[1] + q1, l);
            sxw = f_readlane(S.sin[b].x, l); fs = (int)f_readlane((S.sin[b].y & 0xffffu) + (q1 >> 16), l);
          }
        }
      });
      if (OVF && found < 0 && ngu > 64 * FRB) {                      // reducers 128..: their order is sx[], their data the basis-order arrays
        const int novf = ngu - 64 * FRB;
        for (int k0 = 0; k0 < novf && found < 0; k0 += WAVE) {
          const int k = k0 + lane;
          const int gk = k < novf ? (int)sx[k] : 0;
          const M2 fl = lm[gk];
          const uint64_t mb = ballot64(k < novf && m_divides(fl, hm));
          if (mb) {
            const int l = __builtin_ctzll(mb);
            found = 64 * FRB + k0 + l;
            const int gs = (int)f_readlane((uint32_t)gk, l);
            const M2 fm = f_readlane(fl, l), ft = tm[gs];
            const uint2 info = gi[gs];
            const uint32_t tc = info.x >> 16, q1 = h0b - fm.w[1];
            nb0 = (uint32_t)uni((int)(ft.w[0] + (h0a - fm.w[0]))); nb1 = (uint32_t)uni((int)(ft.w[1] + q1));
            sxw = (uint32_t)uni((int)(tc | (negmod(mulmod(tc, info.y & 0xffffu)) << 16)));
            fs = uni((int)((info.y >> 16) + (q1 >> 16)));
          }
        }
      }
      if (found >= 0) {                                              // h <- h - (LT h / LT f) f
        const uint32_t tcg = sxw & 0xffffu, kg = sxw >> 16;          // kg = -tc / lc mod p, formed once when f entered the basis
        hsug = fs > hsug ? fs : hsug;
        if (hsug > 65535) return;                                    // reported by the caller; nothing has been modified
        // the new term b = -(c tc) (tail f * q) takes the place of the cancelled lead term; then (b, h1) are put in
        // order (polynomials.cpp:148-177 on single optional terms) — in the common case nothing moves
        h0c = mulmod(h0c, kg);                                       // -(c_h / lc) tc; 0 when f has no tail
        h0a = nb0; h0b = nb1;
        if (ACCT) bytes += 8 * (found + 1) + 12 * (tcg ? 2 : 1) + 12 * hn;
        if (h0c == 0) { h0c = h1c; h0a = h1a; h0b = h1b; h1c = 0; }
        else if (h1c != 0) {
          const uint64_t kx = (((uint64_t)h1b << 32) | h1a) ^ 0x0000FFFFFFFFFFFFull;
          const uint64_t ky = (((uint64_t)h0b << 32) | h0a) ^ 0x0000FFFFFFFFFFFFull;
          if (kx > ky) {                                             // the old tail leads: exchange
            const uint32_t tc_ = h0c, ta_ = h0a, tb_ = h0b;
            h0c = h1c; h0a = h1a; h0b = h1b; h1c = tc_; h1a = ta_; h1b = tb_;
          } else if (kx == ky) { h0c = addmod(h0c, h1c); h1c = 0; }  // a zero sum drops the term (h becomes 0)
        }
        if (ACCT) bytes += 12 * ((h0c ? 1 : 0) + (h1c ? 1 : 0));
        nred++;                                        // (terminates: the lead monomial strictly decreases)
      } else {                                                       // r <- r + LT h ; h <- h - LT h
        if (ACCT) bytes += 8 * ngu + 12 * (2 * hn - 1);
        if (r0c == 0) { r0c = h0c; r0a = h0a; r0b = h0b; } else { r1c = h0c; r1a = h0a; r1b = h0b; }
        const int d = (int)(h0b >> 16);
        rsug = d > rsug ? d : rsug;
        h0c = h1c; h0a = h1a; h0b = h1b; h1c = 0;
      }
    };
    if constexpr (!ACCT) {
      // The lean variants run this loop as hand-scheduled assembly.  Written in C++ (the loop below, which the accounting
      // variants keep: it is the specification) the compiler carries the joins of its paths as 64-bit flag words and
      // scalar moves — about 58 scalar-pipe instructions per reduction round, on the unit that binds this kernel
      // (DESIGN.md section 4.1); here a round in which the first 64 reducers hold the divisor takes 22.
      // Register roles: h0 = (h0c; h0a, h0b) lead term of h, h1 its tail term (coefficient 0: none), r0 / r1 the remainder,
      // (its monomials are outputs only: whoever reads them looks at the coefficient first), nred / rsug / hsug as in the C++ loop.  Lane l holds reducers l (A) and l + 64 (B): lead monomial (lm0, lm1), tail
      // monomial (tm0, tm1), inx = tail coefficient | (-tc / lc) << 16, sug = sugar.  gfx950 wait states observed: a packed
      // (VOP3P) result needs one state before a VALU reads it (s_nop 0); everything else here is interlocked.
      const int ng_s = ngu;
      uint32_t nred_u = 0, rsug_u = 0, hsug_u = (uint32_t)hsug;
      uint32_t sl, sf, sx_, sxx, sq;
      uint32_t t0, t1, t2;
      // divisibility test of bank P's reducers against LT h; every lane also prepares the new term's monomial (t0, t2) and the
      // sugar candidate (t1) should its reducer be the divisor
#define FA_SCAN(P) \
        "v_pk_sub_u16 %[t0], %[" #P "lm0], %[h0a] clamp\n\t" \
        "v_pk_sub_u16 %[t1], %[" #P "lm1], %[h0b] clamp\n\t" \
        "s_nop 0\n\t" \
        "v_or_b32 %[t0], %[t0], %[t1]\n\t" \
        "v_cmp_eq_u32 vcc, 0, %[t0]\n\t" \
        "v_sub_u32 %[t1], %[h0b], %[" #P "lm1]\n\t" \
        "v_sub_u32 %[t0], %[h0a], %[" #P "lm0]\n\t" \
        "v_add_u32 %[t2], %[t1], %[" #P "tm1]\n\t" \
        "v_add_u32 %[t0], %[t0], %[" #P "tm0]\n\t" \
        "v_lshrrev_b32 %[t1], 16, %[t1]\n\t" \
        "v_add_u32 %[t1], %[t1], %[" #P "sug]\n\t"
      // the first divisor of bank P (vcc non-zero): its quotient data travel by v_readlane
#define FA_PICK(P) \
        "s_ff1_i32_b64 %[sl], vcc\n\t" \
        "v_readlane_b32 %[sf], %[t1], %[sl]\n\t" \
        "v_readlane_b32 %[sx], %[" #P "inx], %[sl]\n\t" \
        "s_max_i32 %[hsug], %[hsug], %[sf]\n\t" \
        "s_cmp_gt_i32 %[hsug], 0xffff\n\t" \
        "s_cbranch_scc1 L_done_%=\n\t" \
        "v_readlane_b32 %[h0a], %[t0], %[sl]\n\t" \
        "v_readlane_b32 %[h0b], %[t2], %[sl]\n\t"
      // a later bank P (reducers 64 b ..): only when the basis reaches into it (NGMIN = 64 b + 1), next bank or L_tm behind it
#define FA_TRY(P, NGMIN, NEXT) \
        "L_try" #P "_%=:\n\t" \
        "s_cmp_lt_i32 %[ng], " #NGMIN "\n\t" \
        "s_cbranch_scc1 L_tm_%=\n\t" \
        FA_SCAN(P) \
        "s_cbranch_vccz " NEXT "_%=\n\t" \
        FA_PICK(P) \
        "s_branch L_red_%=\n\t"
#define FA_HEAD(NEXT) \
        "L_top_%=:\n\t" \
        "s_cmp_eq_u32 %[h0c], 0\n\t" \
        "s_cbranch_scc1 L_done_%=\n\t" \
        FA_SCAN(a) \
        "s_cbranch_vccz " NEXT "_%=\n\t" \
        FA_PICK(a) \
        "L_red_%=:\n\t"                                   /* h <- h - (LT h / LT f) f: the new term takes the lead term's place */ \
        "s_lshr_b32 %[sq], %[sx], 16\n\t" \
        "s_mul_i32 %[sxx], %[h0c], %[sq]\n\t" \
        "s_mul_hi_u32 %[sq], %[sxx], 0x4187a4af\n\t" \
        "s_lshr_b32 %[sq], %[sq], 13\n\t" \
        "s_mul_i32 %[sq], %[sq], 0x7d03\n\t" \
        "s_sub_u32 %[h0c], %[sxx], %[sq]\n\t" \
        "s_add_u32 %[nred], %[nred], 1\n\t" \
        "s_cmp_eq_u32 %[h0c], 0\n\t" \
        "s_cbranch_scc1 L_shift_%=\n\t" \
        "s_cmp_eq_u32 %[h1c], 0\n\t" \
        "s_cbranch_scc1 L_top_%=\n\t" \
        "s_xor_b32 %[sxx], %[h1b], 0xffff\n\t"           /* grevlex keys: high word ^ 0xffff, low word complemented */ \
        "s_xor_b32 %[sq], %[h0b], 0xffff\n\t" \
        "s_cmp_lt_u32 %[sxx], %[sq]\n\t" \
        "s_cbranch_scc1 L_top_%=\n\t"                     /* the new term leads: nothing moves */ \
        "s_cmp_eq_u32 %[sxx], %[sq]\n\t" \
        "s_cbranch_scc0 L_swap_%=\n\t" \
        "s_cmp_lt_u32 %[h1a], %[h0a]\n\t" \
        "s_cbranch_scc1 L_swap_%=\n\t" \
        "s_cmp_eq_u32 %[h1a], %[h0a]\n\t" \
        "s_cbranch_scc0 L_top_%=\n\t" \
        "s_add_u32 %[sxx], %[h0c], %[h1c]\n\t"            /* equal monomials: one term, coefficients added mod p */ \
        "s_add_u32 %[sq], %[sxx], 0xffff82fd\n\t" \
        "s_min_u32 %[h0c], %[sxx], %[sq]\n\t" \
        "s_mov_b32 %[h1c], 0\n\t" \
        "s_branch L_top_%=\n\t" \
        "L_swap_%=:\n\t"                                  /* the old tail leads */ \
        "s_mov_b32 %[sxx], %[h0c]\n\t" "s_mov_b32 %[h0c], %[h1c]\n\t" "s_mov_b32 %[h1c], %[sxx]\n\t" \
        "s_mov_b32 %[sxx], %[h0a]\n\t" "s_mov_b32 %[h0a], %[h1a]\n\t" "s_mov_b32 %[h1a], %[sxx]\n\t" \
        "s_mov_b32 %[sxx], %[h0b]\n\t" "s_mov_b32 %[h0b], %[h1b]\n\t" "s_mov_b32 %[h1b], %[sxx]\n\t" \
        "s_branch L_top_%=\n\t" \
        "L_shift_%=:\n\t"                                 /* the new term vanished (a reducer without tail): h <- its tail term */ \
        "s_mov_b32 %[h0c], %[h1c]\n\t" "s_mov_b32 %[h0a], %[h1a]\n\t" "s_mov_b32 %[h0b], %[h1b]\n\t" "s_mov_b32 %[h1c], 0\n\t" \
        "s_branch L_top_%=\n\t"
#define FA_TAIL \
        "L_tm_%=:\n\t"                                    /* r <- r + LT h ; h <- h - LT h */ \
        "s_cmp_eq_u32 %[r0c], 0\n\t" \
        "s_cbranch_scc0 L_tm1_%=\n\t" \
        "s_mov_b32 %[r0c], %[h0c]\n\t" "s_mov_b32 %[r0a], %[h0a]\n\t" "s_mov_b32 %[r0b], %[h0b]\n\t" \
        "s_branch L_tm2_%=\n\t" \
        "L_tm1_%=:\n\t" \
        "s_mov_b32 %[r1c], %[h0c]\n\t" "s_mov_b32 %[r1a], %[h0a]\n\t" "s_mov_b32 %[r1b], %[h0b]\n\t" \
        "L_tm2_%=:\n\t" \
        "s_lshr_b32 %[sxx], %[h0b], 16\n\t" \
        "s_max_i32 %[rsug], %[rsug], %[sxx]\n\t" \
        "s_mov_b32 %[h0c], %[h1c]\n\t" "s_mov_b32 %[h0a], %[h1a]\n\t" "s_mov_b32 %[h0b], %[h1b]\n\t" "s_mov_b32 %[h1c], 0\n\t" \
        "s_branch L_top_%=\n\t" \
        "L_done_%=:"
#define FA_OUTS \
          [h0c] "+s"(h0c), [h0a] "+s"(h0a), [h0b] "+s"(h0b), [h1c] "+s"(h1c), [h1a] "+s"(h1a), [h1b] "+s"(h1b), \
          [r0c] "+s"(r0c), [r0a] "=&s"(r0a), [r0b] "=&s"(r0b), [r1c] "+s"(r1c), [r1a] "=&s"(r1a), [r1b] "=&s"(r1b), \
          [nred] "+s"(nred_u), [rsug] "+s"(rsug_u), [hsug] "+s"(hsug_u), \
          [sl] "=&s"(sl), [sf] "=&s"(sf), [sx] "=&s"(sx_), [sxx] "=&s"(sxx), [sq] "=&s"(sq), \
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
#define FA_BANK(P, b) \
          [P##lm0] "v"(S.slm[b].w[0]), [P##lm1] "v"(S.slm[b].w[1]), [P##tm0] "v"(S.stm[b].w[0]), [P##tm1] "v"(S.stm[b].w[1]), \
          [P##inx] "v"(S.sin[b].x), [P##sug] "v"(S.sin[b].y & 0xffffu)
      if constexpr (!OVF) {
        asm volatile(FA_HEAD("L_tryb") FA_TRY(b, 65, "L_tm") FA_TAIL
                     : FA_OUTS : FA_BANK(a, 0), FA_BANK(b, 1), [ng] "s"(ng_s) : "vcc", "scc");
      } else {
        // Reducers beyond the registers (|G| > 128: about once per 10^5 episodes of 3-20-10-weighted): at the first round whose
        // lead term no register reducer divides the assembly leaves with bit 31 of the round counter set, and the rest of this
        // reduction runs in round_cpp, which also looks through the overflow order.
        asm volatile(FA_HEAD("L_tryb") FA_TRY(b, 65, "L_ovf")
                     "L_ovf_%=:\n\t"
                     "s_cmp_lt_i32 %[ng], 129\n\t"
                     "s_cbranch_scc1 L_tm_%=\n\t"
                     "s_bitset1_b32 %[nred], 31\n\t"
                     "s_branch L_done_%=\n\t"
                     FA_TAIL
                     : FA_OUTS : FA_BANK(a, 0), FA_BANK(b, 1), [ng] "s"(ng_s) : "vcc", "scc");
      }
#undef FA_SCAN
#undef FA_PICK
#undef FA_TRY
#undef FA_HEAD
#undef FA_TAIL
#undef FA_OUTS
#undef FA_BANK
      nred = (int)nred_u; rsug = (int)rsug_u; hsug = (int)hsug_u;
      if (OVF && __builtin_expect(nred < 0, 0)) {
        nred &= 0x7fffffff;
        while (h0c != 0 && hsug <= 65535) round_cpp(std::true_type{});
      }
    } else
    while (h0c != 0 && hsug <= 65535) round_cpp(std::true_type{});
    if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; break; }
    rsug = rsug > hsug ? rsug : hsug;
    BTerm<2> r0, r1;
    r0.c = r0c; r0.m.w[0] = r0a; r0.m.w[1] = r0b; r1.c = r1c; r1.m.w[0] = r1a; r1.m.w[1] = r1b;

    FSTAMP(3);                                             // 3: reduce
    // ---- basis / pair-set update (buchberger.cpp:321-327) ------------------------------------------------------------
    const int nG_before = nG, nP_before = nP - 1;
    if (uni((int)r0.c) == 0) {                             // zero reduction: only P.erase(remove(action)), stable
      if (nP <= 64) {                                      // all in the register copy: shift it, store the moved part
        const uint32_t sh = f_wave_shl(PA);
        const bool moved = lane >= action && lane < nP - 1;
        PA = lane >= action ? sh : PA;
        if (moved) pairs[lane] = PA;
      } else {
        for (int base = action; base < nP - 1; base += WAVE) {
          const int k = base + lane;
          uint32_t v = 0;
          if (k < nP - 1) v = pairs[k + 1];
          wave_sync();
          if (k < nP - 1) pairs[k] = v;
          wave_sync();
        }
        PA = pairs[lane];
      }
      nP -= 1;
      zero_red++;
    } else {
      // (the common case first: basis means are ~35; the four-bank form only beyond 128 elements, about once per 10^5
      // episodes of 3-20-10-weighted, and said to be unlikely: the register allocator then keeps its copies out of the way)
      if (__builtin_expect(nG < 64, 1)) add_poly(r0, r1, rsug, action, std::integral_constant<int, 1>{});
      else if (NBK == 2 || __builtin_expect(nG < 128, 1)) add_poly(r0, r1, rsug, action, std::integral_constant<int, 2>{});
      else add_poly(r0, r1, rsug, action, std::integral_constant<int, NBK>{});
      if (ACCT) bytes += 12 * (r1.c ? 2 : 1) + 8 * nG_before + 8 * (nP_before + nP);
    }
    FSTAMP(4);                                             // 4: add_poly (pair update, insert)
    if (ACCT) { bytes += nP * obs_row_bytes; bytes_total += bytes; }
    last_nred = vzero + nred;
    if (VAL) value_accumulate(vret, vdisc, p.rewards_mode == BBX_REW_ADDITIONS ? (-1.0 - (double)nred) : -1.0, p.gamma);
    adds += 1 + nred; t_agent++;
#ifndef BBX_PRIO_SHIFT
#define BBX_PRIO_SHIFT 6
#endif
    static_assert(BBX_PRIO_SHIFT == 6, "one test per step serves the hash refill, the priority rotation and the time slice");
    bool slice_over = false;
    if ((t_agent & 63) == 0) {                             // every 64th step: the three things that need no finer grain
      hv = bbx_agent_hash32(agent_seed, (uint32_t)(t_agent + lane));
      if constexpr (PERSIST) {                             // the time slice (see BBX_ST_TIMESLICE): 64 steps are ~0.2 ms of a 10 ms slice
        const uint32_t lim = f_cold_params()->slice_ticks;
        slice_over = lim && (uint32_t)__builtin_amdgcn_s_memrealtime() - t_begin > lim;
      }
#ifndef BBX_NO_PRIO_ROTATION
      // The instruction arbiter serves the OLDEST wave first, and this kernel is bound by the one scalar unit its waves share:
      // left alone, the waves of the workgroups dispatched first run at 2.1 us per step and those dispatched last at 3.8,
      // so every launch ends with a phase in which only the starved quarter is left, too few waves to fill the unit.
      // Rotating the user priority (which ranks above age) every 2^BBX_PRIO_SHIFT steps, offset by the workgroup's quarter of
      // the grid, gives every wave of a SIMD the same share over time: all finish together.  (s_setprio takes an immediate.)
      switch (((t_agent >> BBX_PRIO_SHIFT) + (env >> 10)) & 3) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
      }
#endif
    }
    const bool done = nP == 0;

    if (obs_step) { if (obs32) write_obs32(); else write_obs(true, false); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
    if (TRACE && tracing) {
      const uint64_t oh = write_obs(false, true);
      uint64_t ph = 0;
      for (int r = lane; r < nP; r += WAVE) {
        const uint32_t q = pairs[r];
        ph += bbx_mix64((uint64_t)(2 * r), q & 0xffffu) + bbx_mix64((uint64_t)(2 * r + 1), q >> 16);
      }
      ph = wave_sum64(ph);
      uint64_t nh = 0;
      if (nG > nG_before) {                                          // oracle/trace.py poly_words of the new element
        if (lane == 0) {
          nh = bbx_mix64(0, r1.c ? 2u : 1u) + bbx_mix64(1, r0.c);
          for (int v = 0; v < BBX_MAXVARS; v++) nh += bbx_mix64(2 + v, v < 3 ? m_exp(r0.m, v) : 0u);
          if (r1.c) {
            nh += bbx_mix64(10, r1.c);
            for (int v = 0; v < BBX_MAXVARS; v++) nh += bbx_mix64(11 + v, v < 3 ? m_exp(r1.m, v) : 0u);
          }
        }
        nh = wave_sum64(nh);
      }
      if (lane == 0) {
        BbxTraceRec& tr = p.trace[(size_t)env * p.trace_stride + trace_pos];
        tr.action = action; tr.nP = nP; tr.nG = nG; tr.done = done ? 1 : 0;
        tr.reward = p.rewards_mode == BBX_REW_ADDITIONS ? (-1.0 - (double)nred) : -1.0;
        tr.obs_hash = oh; tr.pairs_hash = ph; tr.newpoly_hash = nh;
      }
    }
    if constexpr (POL > 0) {
      const FColdPolicy pol = f_cold_policy();
      if (pol->post_obs && p.obs) { write_obs32(); obs_trunc |= nP > p.obs_rows ? 1 : 0; }   // (the block a per-step call leaves: the NEW state)
      if (lane == 0) {
        const size_t tb = (size_t)pol_tt * (size_t)pol->stride_out + (size_t)env;
        if (pol->rewards_t) pol->rewards_t[tb] = p.rewards_mode == BBX_REW_ADDITIONS ? (-1.0 - (double)nred) : -1.0;
        if (pol->dones_t) pol->dones_t[tb] = done ? 1 : 0;
        if (pol->post_obs && pol->rows_t) pol->rows_t[env] = nP;
      }
    }
    if constexpr (PERSIST && POL == 0 && !HL) mb_pending = f_cold_params()->mbox != 0;   // (published at the loop top, behind a reset)
    budget--; if (TRACE) trace_pos++;
    done_last = done ? 1 : 0;
    if (done) { episodes++; if (auto_reset) need_reset = 1; }
    if (PERSIST && slice_over && budget > 0) { status = BBX_ST_TIMESLICE; FSTAMP(5); break; }   // steps are still owed: the next kernel takes them
    FSTAMP(5);                                             // 5: observation + bookkeeping
  }
  const FColdParams cz = f_cold_params();
  if (PROF && cz->prof && lane == 0) for (int i = 0; i < 8; i++) cz->prof[(size_t)env * 8 + i] = prof_sum[i];

  const bool handoff = status == BBX_ST_SPILL;
  if (handoff && lane == 0 && cz->ctl_stats) atomicAdd((unsigned long long*)cz->ctl_stats + 1, 1ull);   // statistics: environments that left the class
  if (POL == 0 && p.obs && status == BBX_ST_OK) { if (obs32) write_obs32(); else write_obs(true, false); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
  if (staged_in) {                                                   // write the live prefixes back to the HBM record
    wave_sync();
    F_HBM_PTRS(cz)
    f_banks<NBK>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      const int i = lane + 64 * b;
      if (i < nG) {
        if constexpr (b < FRB) { g_slm[i] = S.slm[b]; g_stm[i] = S.stm[b]; g_si[i] = S.sin[b]; }
        else {                                                       // the record's reducer-order entry, rebuilt from the basis-order arrays
          const int gk = sx[i - 64 * FRB];
          const uint2 info = gi[gk];
          const uint32_t tc = info.x >> 16;
          g_slm[i] = lm[gk]; g_stm[i] = tm[gk];
          g_si[i] = make_uint2(tc | (negmod(mulmod(tc, info.y & 0xffffu)) << 16), (info.y >> 16) | ((uint32_t)gk << 16));
        }
        g_lm[i] = lm[i]; g_tm[i] = tm[i]; g_gi[i] = gi[i];
      }
    });
    for (int i = lane; i < nP; i += WAVE) g_pr[i] = pairs[i];
  }
  if (lane == 0) {
    BbxHdr* h = (BbxHdr*)(cz->recs + (size_t)env * cz->rec_bytes);
    // steps done = the rollout budget this launch started with minus what is left (the header still holds the old one)
    const int budget0 = cz->sess_target ? cz->sess_target - h->sess_done
                                        : (cz->set_budget ? (bbx_st_capacity(status) ? h->budget + cz->nsteps : cz->nsteps) : h->budget);
    const int steps_done = PERSIST ? t_agent - h->t : budget0 - budget;
    if (PERSIST) {
      h->sess_done = (cz->set_budget ? 0 : h->sess_done) + steps_done;
      if (!cz->set_budget && steps_done > 0 && cz->ctl_stats) atomicAdd((unsigned long long*)cz->ctl_stats, (unsigned long long)steps_done);   // statistics: steps taken by later kernels of sessions
    } else if (cz->sess_target) h->sess_done = cz->sess_target - budget;
    rollout_pos = (cz->set_budget ? 0 : h->rollout_pos) + steps_done;
    h->nG = nG; h->nP = nP; h->arena_used = 0; h->status = status; h->need_reset = need_reset;
    h->q_head = q_head; h->t = t_agent; h->total_steps += steps_done; h->total_additions += adds;
    h->gen_rng = gen_state;
    if (!HL) h->std_rng = std_rng;
    if (VAL) { h->vret = vret; h->vdisc = vdisc; if (cz->values) cz->values[env] = vret; }
    h->episodes += episodes; h->zero_reductions += zero_red; h->steps_done = steps_done;
    h->budget = budget; h->rollout_pos = rollout_pos; h->done_last = done_last; h->alg_bytes += bytes_total;
    const int trunc_all = (cz->set_budget ? 0 : h->obs_trunc) | obs_trunc;
    h->obs_trunc = trunc_all;
    if (!handoff) {
      double* rw = cz->rewards; uint8_t* dn = cz->dones; int32_t* rws = cz->rows;
      // (a later kernel of a session that found nothing owed leaves the reward of the step an earlier one took alone)
      if (rw && (steps_done > 0 || (cz->pass == 0 && !cz->sess_target)))
        rw[env] = last_nred < 0 ? 0.0 : (cz->rewards_mode == BBX_REW_ADDITIONS ? (-1.0 - (double)last_nred) : -1.0);
      if (dn) dn[env] = (uint8_t)((done_last || (nP == 0 && !need_reset)) ? 1 : 0);
      if (rws) rws[env] = nP;
    }
    if (cz->lite) {
      // the host may be spinning on this word (done_seq): everything else this wave wrote — rewards, rows, the observation
      // block in host memory — has to be visible first
      // (a mailbox session: the sequence number of the last step this environment took, as its per-step publication left it)
      const int seq = (PERSIST && cz->mbox) ? (h->sess_done % 16000) + 1 : cz->done_seq;
      int32_t* lw = cz->lite + 4 * (size_t)env;
      const int word0 = status | (trunc_all ? BBX_LITE_OBS_TRUNC : 0) | (seq << 17);
      if (seq) {                                       // the word the host watches goes last and alone, behind a system-scope fence
        // (a mailbox session: bit 30 of the budget word says "this wave has left and stored its environment" — what the host
        // waits for when it closes the session, instead of the runtime's completion signal)
        lw[1] = q_head; lw[2] = budget; lw[3] = nP;
        __threadfence_system();
        __hip_atomic_store(lw, word0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // (the mark is this wave's LAST store: the host, seeing it, clears the status words and may begin the next session at once —
        // a status word arriving behind the mark carried this session's last sequence number into the next one, where a session of
        // one step has the number the next session's first step waits for: the host then returned the OLD step's outputs.  Found
        // by scripts/fuzz_gym.py mixing bbx_step_obs and bbx_step calls.)
        if (PERSIST && cz->mbox) __hip_atomic_store(lw + 2, budget | 0x40000000, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else *(int4*)lw = make_int4(word0, q_head, budget, nP);
    }
  }
}

template <bool TRACE, bool ACCT>
__global__ __launch_bounds__(256, 4) void bbx_fast_kernel(BbxFastParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<TRACE, ACCT>(p, smem);
}
__global__ __launch_bounds__(256, 4) void bbx_fast_headline_kernel(BbxFastParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<false, false, false, true>(p, smem);
}
// value() rollouts of the register/LDS-resident class
__global__ __launch_bounds__(256, 4) void bbx_fast_value_kernel(BbxFastParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<false, false, false, false, 0, false, true>(p, smem);
}
// bbx_policy_step_device calls served by a persistent session (fast_body POL + PERSIST)
template <int NB>
__global__ __launch_bounds__(256, 4) void bbx_fast_policy_session_kernel(BbxFastPolicyParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<false, false, false, false, NB, true>(q.f, smem);
}
// the kernels of persistent sessions (fast_body PERSIST): the headline shape and the general lean one
__global__ __launch_bounds__(256, 4) void bbx_fast_headline_persistent_kernel(BbxFastParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<false, false, false, true, 0, true>(p, smem);
}
__global__ __launch_bounds__(256, 4) void bbx_fast_persistent_kernel(BbxFastParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<false, false, false, false, 0, true>(p, smem);
}
// Policy + step in one launch: every wave first evaluates the PMLP policy on its environment's rows of the
// observation block the previous launch left (pmlp_act_wave: matrix-core hidden layer, log-softmax, inverse-CDF draw),
// then takes the step with the sampled row as its action.  The two phases use the same registers and the same LDS one
// after the other; a vector step costs one kernel's launch, ramp and drain instead of two.  The struct starts with the
// step parameters: f_cold_params() reads them at offset 0 of the kernel arguments.
template <int NB, int KS>
__global__ __launch_bounds__(256, 4) void bbx_fast_policy_kernel(BbxFastPolicyParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int env = blockIdx.x * (blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
  const int action = pmlp_act_wave<NB, KS>(smem, env, env < q.f.B, q.f.obs, q.f.rows, q.f.obs_rows, 2 * q.f.k * q.f.nvars, q.pol.wp, q.pol.u,
                                           q.pol.actions, q.pol.logprobs);
  __syncthreads();                                     // the policy's LDS scratch becomes the step's state
  fast_body<false, false, false, false, 0, false, false, FNBK_POL>(q.f, smem, action);   // (the policy's registers and LDS leave room for two banks)
}
// policy rollout: nsteps steps per launch with the policy inside the step loop (fast_body POL)
template <int NB>
__global__ __launch_bounds__(256, 4) void bbx_fast_policy_rollout_kernel(BbxFastPolicyParams q) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<false, false, false, false, NB>(q.f, smem);
}
#ifdef BBX_PROF_BUILD
// diagnostic build with s_memtime stamps between the phases of a step (never timed, never shipped as a result)
__global__ __launch_bounds__(256, 4) void bbx_fast_prof_kernel(BbxFastParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  fast_body<false, false, true>(p, smem);
}
#endif
