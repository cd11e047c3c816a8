// General class of the step kernel: term arena + scratch polynomials in HBM/L2, merge-path tiles through per-wave LDS
// (non-binomial random ideals, 8-variable fixed ideals, big batches of fixed ideals).  One wavefront per environment.
// Reference semantics: buchberger.cpp:18-99, 299-329, 354-408; polynomials.cpp:41-202.
#include "bbx_device.h"

// ------------------------------------------------------------------ the step kernel
template <int W, bool STAGED, bool TRACE, bool PROF = false>
__device__ __forceinline__ void step_body(const BbxParams& p, char* smem, unsigned long long* prof_out = nullptr) {
  const int lane = lane_id();
  unsigned long long ps[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // diagnostic build: cycles per phase
  unsigned long long pl = PROF ? __builtin_amdgcn_s_memtime() : 0;
#define GSTAMP(slot) do { if (PROF) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); ps[slot] += t_ - pl; pl = t_; } } while (0)
  const int wave_in_block = uni((int)(threadIdx.x / WAVE));   // provably wave-uniform: record addresses live in SGPRs
  const int env = (int)(blockIdx.x * (blockDim.x / WAVE) + wave_in_block);
  if (env >= p.B) return;                       // whole wave exits together
  char* grec = p.recs + (size_t)env * p.L.rec_bytes;
  BbxHdr* ghdr = (BbxHdr*)grec;
  const BbxLayout& L = STAGED ? p.LL : p.L;      // the layout this kernel works in

  int nG = uni(ghdr->nG), nP = uni(ghdr->nP), arena_used = uni(ghdr->arena_used);
  int status = uni(ghdr->status), need_reset = uni(ghdr->need_reset), q_head = uni(ghdr->q_head);
  int t_agent = uni(ghdr->t), episode_steps = uni(ghdr->episode_steps);
  int episodes = uni(ghdr->episodes), zero_red = uni(ghdr->zero_reductions);
  long long total_steps = ghdr->total_steps, total_adds = ghdr->total_additions, alg_bytes = ghdr->alg_bytes;
  const uint32_t agent_seed = uni((int)ghdr->agent_seed);
  uint32_t std_rng = (uint32_t)uni((int)ghdr->std_rng);
  uint32_t gen_state = ghdr->gen_rng;
  int budget = uni(ghdr->budget), rollout_pos = uni(ghdr->rollout_pos);
  int done_last = uni(ghdr->done_last);
  if (status == BBX_ST_STARVED || status == BBX_ST_SPILL || status == BBX_ST_TIMESLICE) status = BBX_ST_OK;   // transient states: try again
  double vret = ghdr->vret, vdisc = ghdr->vdisc;
  int obs_trunc = uni(ghdr->obs_trunc);
  // (an environment waiting for the host to enlarge its record keeps the steps it still owes: bbx_common.h)
  if (p.set_budget) { budget = bbx_st_capacity(status) ? budget + p.nsteps : p.nsteps; rollout_pos = 0; done_last = 0; vret = 0.0; vdisc = 1.0; obs_trunc = 0; }
  if (p.pass == 1 && !(status == BBX_ST_OK && (need_reset || (budget > 0 && nP > 0)))) return;  // nothing left to do here

  Env<W> ge = env_view<W>(grec, p.L);
  // In the staged instantiation the working view is ALWAYS the LDS copy (never a select between an LDS and
  // a global pointer), so that every access below compiles to ds_read/ds_write instead of flat_*.
  Env<W> e = STAGED ? env_view<W>(smem + (size_t)wave_in_block * L.rec_bytes, L) : ge;
  bool staged_in = false;
  if (STAGED) {
    if (status == BBX_ST_OK && !(!need_reset && nP == 0)) {   // (an idle environment is not staged: bbx_fast.h, idle0)
      if (nG > (int)L.maxG || nP > (int)L.maxP || arena_used > (int)L.arena) status = BBX_ST_SPILL;
      else {
        stage_copy<W>(e, ge, nG, nP, arena_used);
        staged_in = true;
        wave_sync();
      }
    }
  }
  int steps_done = 0;
  double last_reward = 0.0;
  const bool tracing = TRACE && p.trace != nullptr;   // hashing code exists only in the TRACE instantiations
  // per-wave LDS tile scratch of the merge-path merge (HBM-resident class only; the launcher provides it)
  char* mlds = (!STAGED && smem != nullptr) ? smem + (size_t)wave_in_block * merge_lds_bytes<W>() : nullptr;

  // scratch polynomials
  const int maxT = (int)L.maxT;
  Mono<W>* hm0 = e.hm;            uint16_t* hc0 = e.hc;
  Mono<W>* hm1 = e.hm + maxT;     uint16_t* hc1 = e.hc + maxT;
  Mono<W>* rm = e.hm + 2 * maxT;  uint16_t* rc = e.hc + 2 * maxT;
  Mono<W>* tm = e.hm + 3 * maxT;  uint16_t* tc = e.hc + 3 * maxT;   // 2*maxT staging

  uint32_t rng_mark = std_rng;                  // the selection engine's state before the step in progress
  for (;;) {
    if (status != BBX_ST_OK) break;
    rng_mark = std_rng;
    if (need_reset) {                           // also serves a reset left pending by the last step
      if (!wave_reset<W>(e, p, L, env, nG, nP, arena_used, q_head, &status, gen_state)) {
        // the reset restarts from the same queued ideal: in the LDS class a capacity miss is only a spill
        if (STAGED && (status == BBX_ST_G_FULL || status == BBX_ST_P_FULL || status == BBX_ST_ARENA_FULL)) {
          status = BBX_ST_SPILL; nG = 0; nP = 0; arena_used = 0;
        }
        break;
      }
      need_reset = 0; episode_steps = 0;
    }
    if (budget <= 0) break;
    if (nP == 0) break;                         // finished episode and no auto-reset: nothing to do
    // headroom for the worst case of this step, checked BEFORE anything is modified so that a miss leaves a
    // consistent state: one new basis element of <= maxT terms and at most |G| new pairs
    if (nG + 1 > (int)L.maxG || nP - 1 + nG > (int)L.maxP || arena_used + maxT > (int)L.arena) {
      status = STAGED ? BBX_ST_SPILL : (nG + 1 > (int)L.maxG ? BBX_ST_G_FULL : (nP - 1 + nG > (int)L.maxP ? BBX_ST_P_FULL : BBX_ST_ARENA_FULL));
      break;
    }

    // ---- choose the pair ------------------------------------------------------------------
    int action;
    if (p.agent == BBX_AGENT_EXTERNAL) action = p.actions[env];
    else if (p.agent == BBX_AGENT_HASH) action = (int)bbx_agent_action32(agent_seed, (uint32_t)t_agent, (uint32_t)nP);
    else if (p.agent == BBX_AGENT_FIRST) action = 0;
    else if (p.agent == BBX_AGENT_LAST) action = nP - 1;
    else if (p.agent == BBX_AGENT_STDRANDOM) action = std_choice(std_rng, nP);
    else action = select_pair<W>(e, nP, p.agent, [&](int g) { return (int)e.psug[g]; });
    action = uni(action);
    if (action < 0 || action >= nP) { status = BBX_ST_BAD_ACTION; break; }
    const uint32_t pr = (uint32_t)uni((int)e.pairs[action]);
    const int gi = pr & 0xffffu, gj = pr >> 16;
    // (the pair leaves P — buchberger.cpp:319 — only once the reduction is through: S-polynomial and reduction work in
    // scratch, so a polynomial that outgrows max_poly_terms leaves the record exactly as the step found it and the step
    // is taken again after the host has enlarged the scratch: bbx_common.h, bbx_st_capacity)
    long long sb = 0;                            // algorithmic bytes of this step

    // ---- S-polynomial  buchberger.cpp:18-21 -----------------------------------------------------
    int hn, hoff = 0, hsug;
    Mono<W>* hm = hm0; uint16_t* hc = hc0;
    {
      const Mono<W> lmi = e.lm[gi], lmj = e.lm[gj];
      const Mono<W> gamma = m_lcm(lmi, lmj);
      const int offi = uni((int)e.poff[gi]), offj = uni((int)e.poff[gj]);
      PView<W> A, Bv;
      A.m = e.am + offi + 1; A.c = e.ac + offi + 1; A.n = uni((int)e.plen[gi]) - 1;
      A.shift = m_div(gamma, lmi); A.scale = (uint32_t)uni((int)e.pinv[gi]);
      Bv.m = e.am + offj + 1; Bv.c = e.ac + offj + 1; Bv.n = uni((int)e.plen[gj]) - 1;
      Bv.shift = m_div(gamma, lmj); Bv.scale = negmod((uint32_t)uni((int)e.pinv[gj]));
      int si = uni((int)e.psug[gi]) + (int)m_deg(A.shift), sj = uni((int)e.psug[gj]) + (int)m_deg(Bv.shift);
      hsug = uni(si > sj ? si : sj);
      if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; break; }
      if (!STAGED && p.spill_terms && A.n + Bv.n > p.spill_terms) { status = BBX_ST_SPILL; break; }   // long polynomials: a workgroup's job
      if (A.n + Bv.n > 2 * maxT) { status = BBX_ST_POLY_TOO_LONG; break; }
      GSTAMP(0);                                   // 0: loop top, agent, pair removal
      const bool big = mlds && A.n > 0 && Bv.n > 0 && A.n + Bv.n > 64;
      hn = big ? wave_merge_tiled<W>(A, Bv, mlds, hm, hc, maxT) : wave_merge<W>(A, Bv, tm, tc, hm, hc, maxT);
      if (hn < 0) { status = BBX_ST_POLY_TOO_LONG; break; }
      GSTAMP(1);                                   // 1: S-polynomial merge
      sb += 12LL * (A.n + Bv.n + 2 + hn);          // both inputs read, S-polynomial written
    }

    // ---- reduce  buchberger.cpp:24-49 -----------------------------------------------------------
    int nsteps_red = 0, rn = 0, rsug = 0;
    bool overflow = false;
    while (hn - hoff > 0) {
      const Mono<W> lmh = hm[hoff];
      int found = -1;
      for (int base = 0; base < nG; base += WAVE) {   // first reducer (in G_ order) whose LM divides LM(h)
        int k = base + lane;
        bool d = k < nG && m_divides(e.slm[k], lmh);
        uint64_t mask = ballot64(d);
        if (mask) { found = base + __builtin_ctzll(mask); break; }
      }
      GSTAMP(2);                                  // 2: divisor scans
      if (found >= 0) {                         // h <- h - (LT h / LT f) f     (34-36)
        const int g = uni((int)e.sidx[found]);
        const uint32_t c = mulmod((uint32_t)uni((int)hc[hoff]), (uint32_t)uni((int)e.pinv[g]));
        const int offg = uni((int)e.poff[g]);
        PView<W> A, Bv;
        A.m = hm + hoff + 1; A.c = hc + hoff + 1; A.n = hn - hoff - 1; A.shift = m_zero<W>(); A.scale = 1;
        Bv.m = e.am + offg + 1; Bv.c = e.ac + offg + 1; Bv.n = uni((int)e.plen[g]) - 1;
        Bv.shift = m_div(lmh, e.lm[g]); Bv.scale = negmod(c);
        int fs = uni((int)e.psug[g]) + (int)m_deg(Bv.shift);
        hsug = uni(fs > hsug ? fs : hsug);
        if (hsug > 65535) { status = BBX_ST_DEG_OVERFLOW; overflow = true; break; }
        if (!STAGED && p.spill_terms && A.n + Bv.n > p.spill_terms) { status = BBX_ST_SPILL; overflow = true; break; }
        if (A.n + Bv.n > 2 * maxT) { status = BBX_ST_POLY_TOO_LONG; overflow = true; break; }
        Mono<W>* nm = (hm == hm0) ? hm1 : hm0; uint16_t* nc = (hc == hc0) ? hc1 : hc0;
        GSTAMP(3);                                // 3: reducer fetch / setup
        const bool big = mlds && A.n > 0 && Bv.n > 0 && A.n + Bv.n > 64;
        const int nn = big ? wave_merge_tiled<W>(A, Bv, mlds, nm, nc, maxT) : wave_merge<W>(A, Bv, tm, tc, nm, nc, maxT, PROF ? &ps[6] : nullptr);
        if (nn < 0) { status = BBX_ST_POLY_TOO_LONG; overflow = true; break; }
        GSTAMP(4);                                // 4: reduction merges (6/7: their pass 1 / pass 2)
        sb += 8LL * (found + 1) + 12LL * (Bv.n + 1) + 12LL * (A.n + 1 + nn);
        hm = nm; hc = nc; hn = nn; hoff = 0;
        nsteps_red++;
        if (nsteps_red > (1 << 24)) { status = BBX_ST_RUNAWAY; overflow = true; break; }
      } else {                                  // r <- r + LT h ; h <- h - LT h   (41-44)
        if (rn >= maxT) { status = BBX_ST_POLY_TOO_LONG; overflow = true; break; }
        sb += 8LL * nG + 12LL * (2 * (hn - hoff) - 1);
        if (lane == 0) { rm[rn] = lmh; rc[rn] = hc[hoff]; }
        int d = uni((int)m_deg(lmh));
        rsug = d > rsug ? d : rsug;
        rn++; hoff++;
        GSTAMP(5);                                // 5: tail moves
      }
    }
    if (overflow) break;
    wave_sync();
    rsug = rsug > hsug ? rsug : hsug;            // sugar of r + h (48), h's sugar survives its terms
    if (rn > 65535) { status = BBX_ST_POLY_LIMIT; break; }   // plen[] is 16 bits

    // ---- P.erase(remove(action))  buchberger.cpp:319 — stable; from here on the step cannot fail for capacity -------
    for (int base = action; base < nP - 1; base += WAVE) {
      int k = base + lane;
      uint32_t v = 0;
      if (k < nP - 1) v = e.pairs[k + 1];
      wave_sync();
      if (k < nP - 1) e.pairs[k] = v;
      wave_sync();
    }
    nP -= 1;

    // ---- basis / pair-set update  buchberger.cpp:321-327 ---------------------------------------
    const int nG_before = nG, nP_before = nP;
    if (rn != 0) {
      if (!wave_add_poly<W>(e, L, nG, nP, arena_used, rm, rc, rn, rsug, p.elim, p.sort_reducers, &status)) break;
      sb += 12LL * rn + 8LL * nG_before + 8LL * (nP_before + nP);
    } else zero_red++;
    sb += 4LL * nP * 2 * p.nvars * p.k;             // the observation matrix of the new state
    alg_bytes += sb;
    const double reward = (p.rewards_mode == BBX_REW_ADDITIONS) ? (-1.0 - (double)nsteps_red) : -1.0;  // 328
    last_reward = reward;
    if (p.value_mode) value_accumulate(vret, vdisc, reward, p.gamma);
    total_steps++; total_adds += 1 + nsteps_red; t_agent++; episode_steps++; steps_done++;
    const bool done = nP == 0;

    // ---- the observation a policy would consume after this step ---------------------------------
    if (p.obs_every_step && p.obs) { wave_obs<W>(e, p, env, nP, true, false); obs_trunc |= nP > p.obs_rows ? 1 : 0; }
    // ---- parity trace (tests): hashes of the post-step observation / pair set / new element ---
    if (TRACE && tracing) {
      uint64_t oh = wave_obs<W, true>(e, p, env, nP, false, true);
      uint64_t ph = wave_pairs_hash<W, Env<W>>(e, nP);
      uint64_t nh = nG > nG_before ? wave_poly_hash<W>(e, nG - 1) : 0;
      if (lane == 0) {
        BbxTraceRec& tr = p.trace[(size_t)env * p.trace_stride + rollout_pos];
        tr.action = action; tr.nP = nP; tr.nG = nG; tr.done = done ? 1 : 0; tr.reward = reward;
        tr.obs_hash = oh; tr.pairs_hash = ph; tr.newpoly_hash = nh;
      }
    }
    budget--; rollout_pos++;
    done_last = done ? 1 : 0;
    if (done) {
      episodes++;
      if (p.auto_reset) need_reset = 1;
    }
  }

  if (bbx_st_capacity(status) || status == BBX_ST_SPILL) std_rng = rng_mark;   // the step did not happen: its draw is taken again
  if (PROF && prof_out && lane == 0) for (int i = 0; i < 10; i++) prof_out[(size_t)env * 10 + i] = ps[i];
  // an environment that must continue in the follow-up pass reports nothing yet
  const bool handoff = status == BBX_ST_SPILL;
  // ---- observation of the state the caller sees next ------------------------------------------
  if (p.obs && status == BBX_ST_OK) { wave_obs<W>(e, p, env, nP, true, false); obs_trunc |= nP > p.obs_rows ? 1 : 0; }

  if (STAGED && staged_in) {
    wave_sync();
    stage_copy<W>(ge, e, nG, nP, arena_used);
  }
  if (lane == 0) {
    BbxHdr* h = ghdr;
    h->nG = nG; h->nP = nP; h->arena_used = arena_used; h->status = status; h->need_reset = need_reset;
    h->q_head = q_head; h->t = t_agent; h->std_rng = std_rng; h->gen_rng = gen_state; h->episode_steps = episode_steps; h->total_steps = total_steps;
    h->total_additions = total_adds; h->episodes = episodes; h->zero_reductions = zero_red; h->steps_done = steps_done;
    h->budget = budget; h->rollout_pos = rollout_pos; h->done_last = done_last; h->alg_bytes = alg_bytes;
    h->vret = vret; h->vdisc = vdisc; h->obs_trunc = obs_trunc;
    if (p.lite) *(int4*)(p.lite + 4 * (size_t)env) = make_int4(status | (obs_trunc ? BBX_LITE_OBS_TRUNC : 0), q_head, budget, nP);
    if (p.value_mode && p.values) p.values[env] = vret;
    if (!handoff) {
      if (p.rewards && (steps_done > 0 || p.pass == 0)) p.rewards[env] = last_reward;
      if (p.dones) p.dones[env] = (uint8_t)((done_last || (nP == 0 && !need_reset)) ? 1 : 0);
      if (p.rows) p.rows[env] = nP;
    }
  }
}

template <int W, bool STAGED, bool TRACE>
__global__ __launch_bounds__(256) void bbx_step_kernel(BbxParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  step_body<W, STAGED, TRACE>(p, smem);
}
// the same body under its own name for launches that only reset / refresh observations (nsteps == 0), so that
// profiles of bbx_step_kernel contain step launches only
template <int W>
__global__ __launch_bounds__(256) void bbx_aux_kernel(BbxParams p) {
  step_body<W, false, false>(p, nullptr);
}
#ifdef BBX_PROF_BUILD
// diagnostic build with s_memtime stamps (BBX_PROF=1), never used for reported numbers
template <int W>
__global__ __launch_bounds__(256) void bbx_step_prof_kernel(BbxParams p, unsigned long long* prof) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  step_body<W, false, false, true>(p, smem, prof);
}

#endif

// kind: 0 = HBM-resident step kernel, 1 = LDS-staged step kernel, 2 = aux (reset / observation only)
#define BBX_LAUNCH(KERN) hipLaunchKernelGGL((KERN), dim3(blocks), dim3(threads), lds, stream, *p)
template <int W>
static int launch_general_w(const BbxParams* p, int kind, int blocks, int threads, size_t lds, hipStream_t stream) {
  const bool trace = p->trace != nullptr;
  if (kind == 2) { BBX_LAUNCH(bbx_aux_kernel<W>); return 0; }
  if (kind == 1) {
    const void* fn = trace ? (const void*)bbx_step_kernel<W, true, true> : (const void*)bbx_step_kernel<W, true, false>;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    if (trace) BBX_LAUNCH((bbx_step_kernel<W, true, true>)); else BBX_LAUNCH((bbx_step_kernel<W, true, false>));
    return 0;
  }
#ifdef BBX_PROF_BUILD   // diagnostic build only (-DBBX_PROF_BUILD): per-phase s_memtime sums, never in the product library
  if (!trace && getenv("BBX_PROF")) {
    static unsigned long long* d_prof = nullptr;
    if (!d_prof) (void)hipMalloc((void**)&d_prof, (size_t)p->B * 10 * sizeof(unsigned long long));
    lds = (size_t)(threads / WAVE) * merge_lds_bytes<W>();
    (void)hipFuncSetAttribute((const void*)bbx_step_prof_kernel<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((bbx_step_prof_kernel<W>), dim3(blocks), dim3(threads), lds, stream, *p, d_prof);
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned long long> h((size_t)p->B * 10);
    (void)hipMemcpy(h.data(), d_prof, h.size() * 8, hipMemcpyDeviceToHost);
    double s[10] = {0}, tot = 0;
    for (int e = 0; e < p->B; e++) for (int i = 0; i < 10; i++) s[i] += (double)h[(size_t)e * 10 + i];
    for (int i = 0; i < 6; i++) tot += s[i];
    fprintf(stderr, "[bbx prof general] nsteps=%d kcycles/env:", p->nsteps);
    for (int i = 0; i < 10; i++) fprintf(stderr, " p%d=%.0f(%.0f%%)", i, s[i] / p->B / 1e3, 100.0 * s[i] / tot);
    fprintf(stderr, "\n");
    return 0;
  }
#endif
  lds = (size_t)(threads / WAVE) * merge_lds_bytes<W>();          // merge-path tile scratch, one per wave
  const void* fn = trace ? (const void*)bbx_step_kernel<W, false, true> : (const void*)bbx_step_kernel<W, false, false>;
  hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (err != hipSuccess) return (int)err;
  if (trace) BBX_LAUNCH((bbx_step_kernel<W, false, true>)); else BBX_LAUNCH((bbx_step_kernel<W, false, false>));
  return 0;
}
extern "C" int bbx_launch_general(const BbxParams* p, int kind, int blocks, int threads, size_t lds, hipStream_t stream) {
  return p->L.W == 2 ? launch_general_w<2>(p, kind, blocks, threads, lds, stream)
       : p->L.W == 4 ? launch_general_w<4>(p, kind, blocks, threads, lds, stream) : launch_general_w<8>(p, kind, blocks, threads, lds, stream);
}
