// Hand-tuned register/LDS-resident kernel of the headline class (bbx_fast.h): launcher.
#include "bbx_device.h"
#include "bbx_pmlp.h"
#include "bbx_binom.h"
#include "bbx_fast.h"

// kind 3: the hand-tuned LDS/register-resident kernel (bbx_fast.h) for W == 2 binomial, GM, sorted reducers
extern "C" int bbx_launch_fast(const BbxParams* p, int blocks, int threads, int envs_per_block, hipStream_t stream) {
  BbxFastParams f{};
  f.recs = p->recs; f.qwords = p->q.words; f.qtail = p->q.tail; f.inv_table = p->inv_table;
  f.actions = p->actions; f.rewards = p->rewards; f.dones = p->dones; f.rows = p->rows; f.obs = p->obs; f.trace = p->trace;
  f.rec_bytes = p->L.rec_bytes; f.hbmG = p->L.maxG;
  f.q_env_stride = p->q.env_stride; f.q_slot_words = p->q.slot_words; f.q_nslots = p->q.nslots; f.q_fixed = p->q.fixed;
  f.B = p->B; f.nsteps = p->nsteps; f.obs_rows = p->obs_rows; f.trace_stride = p->trace_stride; f.k = p->k; f.nvars = p->nvars;
  f.lim_G = p->fast_G; f.lim_P = p->fast_P;
  f.agent = p->agent; f.auto_reset = p->auto_reset; f.set_budget = p->set_budget; f.pass = p->pass;
  f.obs_every_step = p->obs_every_step; f.obs_fill = p->obs_fill; f.rewards_mode = p->rewards_mode;
  f.lite = p->lite; f.done_seq = p->done_seq;
  f.gen = p->gen;
  f.sort_input = p->sort_input;
  f.ctl = p->ctl; f.sess_target = p->sess_target; f.ctl_stats = p->ctl_stats; f.slice_ticks = p->slice_ticks; f.mbox = p->mbox;
  f.gamma = p->gamma; f.values = p->values;
  const size_t lds = (size_t)envs_per_block * FLay<FNBK_WIDE>::BYTES, lds_pol = (size_t)envs_per_block * FLay<FNBK_POL>::BYTES;
#ifdef BBX_PROF_BUILD
  static unsigned long long* d_prof = nullptr;
  if (getenv("BBX_PROF") && !p->trace) {          // diagnostic: per-phase cycle sums, printed by bbx_prof_dump()
    if (!d_prof) (void)hipMalloc((void**)&d_prof, (size_t)p->B * 8 * sizeof(unsigned long long));
    f.prof = d_prof;
    hipLaunchKernelGGL(bbx_fast_prof_kernel, dim3(blocks), dim3(threads), lds, stream, f);
    (void)hipStreamSynchronize(stream);
    std::vector<unsigned long long> h((size_t)p->B * 8);
    (void)hipMemcpy(h.data(), d_prof, h.size() * 8, hipMemcpyDeviceToHost);
    double s[8] = {0}; for (int e = 0; e < p->B; e++) for (int i = 0; i < 8; i++) s[i] += (double)h[(size_t)e * 8 + i];
    double tot = 0; for (int i = 0; i < 8; i++) tot += s[i];
    fprintf(stderr, "[bbx prof] nsteps=%d ticks/step/env:", p->nsteps);
    for (int i = 0; i < 6; i++) fprintf(stderr, " p%d=%.0f(%.0f%%)", i, s[i] / p->B / (p->nsteps ? p->nsteps : 1), 100.0 * s[i] / tot);
    fprintf(stderr, "\n");
    return 0;
  }
#endif
  if (p->policy) {                                         // policy + step in one launch (bbx_api.cpp checked the shapes)
    BbxFastPolicyParams q; q.f = f; q.pol = *p->policy;
    q.f.agent = BBX_AGENT_EXTERNAL; q.f.actions = q.pol.actions;
    if (q.pol.rollout) {                                   // nsteps steps, the policy inside the step loop (3 variables, k = 2)
      q.f.actions = nullptr; q.f.rewards = nullptr; q.f.dones = nullptr; q.f.rows = q.pol.post_obs ? q.pol.rows_t : nullptr; q.f.obs_every_step = 0; q.f.auto_reset = 1;
      const size_t rl = (size_t)envs_per_block * (FLay<FNBK_POL>::BYTES + 4 * FLay<FNBK_POL>::P) + ((size_t)(2 * 6 + 2) * 32 * pmlp_nb_for(q.pol.hidden) + 4) * sizeof(float);
      if (p->ctl) {                                        // per-step calls served by a persistent session
        if (pmlp_nb_for(q.pol.hidden) == 2) hipLaunchKernelGGL((bbx_fast_policy_session_kernel<2>), dim3(blocks), dim3(threads), rl, stream, q);
        else hipLaunchKernelGGL((bbx_fast_policy_session_kernel<4>), dim3(blocks), dim3(threads), rl, stream, q);
        return 0;
      }
      if (pmlp_nb_for(q.pol.hidden) == 2) hipLaunchKernelGGL((bbx_fast_policy_rollout_kernel<2>), dim3(blocks), dim3(threads), rl, stream, q);
      else hipLaunchKernelGGL((bbx_fast_policy_rollout_kernel<4>), dim3(blocks), dim3(threads), rl, stream, q);
      return 0;
    }
    const int nb = pmlp_nb_for(q.pol.hidden), ks = pmlp_ks_for(2 * f.k * f.nvars);
    const size_t pl = pmlp_lds_bytes(envs_per_block, f.obs_rows), ll = pl > lds_pol ? pl : lds_pol;
    if (ks == 3) { if (nb == 2) hipLaunchKernelGGL((bbx_fast_policy_kernel<2, 3>), dim3(blocks), dim3(threads), ll, stream, q);
                   else hipLaunchKernelGGL((bbx_fast_policy_kernel<4, 3>), dim3(blocks), dim3(threads), ll, stream, q); }
    else { if (nb == 2) hipLaunchKernelGGL((bbx_fast_policy_kernel<2, 6>), dim3(blocks), dim3(threads), ll, stream, q);
           else hipLaunchKernelGGL((bbx_fast_policy_kernel<4, 6>), dim3(blocks), dim3(threads), ll, stream, q); }
    return 0;
  }
  if (p->value_mode) { hipLaunchKernelGGL(bbx_fast_value_kernel, dim3(blocks), dim3(threads), lds, stream, f); return 0; }   // (lean, untraced: bbx_api.cpp)
  if (p->ctl) {                                            // the kernel of a persistent session (bbx_api.cpp admits lean, untraced launches only)
    if (f.agent == BBX_AGENT_HASH && f.nvars == 3 && f.k == 2 && f.obs && f.obs_every_step && !f.obs_fill && f.auto_reset)
      hipLaunchKernelGGL(bbx_fast_headline_persistent_kernel, dim3(blocks), dim3(threads), lds, stream, f);
    else hipLaunchKernelGGL(bbx_fast_persistent_kernel, dim3(blocks), dim3(threads), lds, stream, f);
    return 0;
  }
  if (p->trace) hipLaunchKernelGGL((bbx_fast_kernel<true, true>), dim3(blocks), dim3(threads), lds, stream, f);
  else if (p->accounting) hipLaunchKernelGGL((bbx_fast_kernel<false, true>), dim3(blocks), dim3(threads), lds, stream, f);
  else if (f.agent == BBX_AGENT_HASH && f.nvars == 3 && f.k == 2 && f.obs && f.obs_every_step && !f.obs_fill && f.auto_reset)
    hipLaunchKernelGGL(bbx_fast_headline_kernel, dim3(blocks), dim3(threads), lds, stream, f);
  else hipLaunchKernelGGL((bbx_fast_kernel<false, false>), dim3(blocks), dim3(threads), lds, stream, f);
  return 0;
}
