// Binomial class of the step kernel (bbx_binom.h): launcher.  kind: 0 = HBM-resident, 1 = LDS-staged, 2 = aux.
#include "bbx_device.h"
#include "bbx_pmlp.h"
#include "bbx_binom.h"

#define BBX_LAUNCH(KERN) hipLaunchKernelGGL((KERN), dim3(blocks), dim3(threads), lds, stream, *p)
template <int W>
static int launch_binom_w(const BbxParams* p, int kind, int blocks, int threads, size_t lds, hipStream_t stream) {
  const bool trace = p->trace != nullptr;
  if (kind == 2) { BBX_LAUNCH(bbx_binom_aux_kernel<W>); return 0; }
  if (kind == 1) {
    const void* fn = trace ? (const void*)bbx_binom_kernel<W, true, true> : (const void*)bbx_binom_kernel<W, true, false>;
    hipError_t err = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (err != hipSuccess) return (int)err;
    if (trace) BBX_LAUNCH((bbx_binom_kernel<W, true, true>)); else BBX_LAUNCH((bbx_binom_kernel<W, true, false>));
    return 0;
  }
    lds = (size_t)(threads / WAVE) * update_lds_bytes<W>();           // Gebauer-Moeller peel scratch, one per wave
    const size_t lds_plain = W == 4 ? (size_t)(threads / WAVE) * BC_BYTES : lds;   // 16-byte monomials: the LDS copy instead (BEnvC)
    if constexpr (W == 2 || W == 4) {
      if (p->policy && p->policy->rollout) {               // a policy rollout: its continuation pass, or the whole of it
        BbxParams q = *p; q.policy = nullptr; q.actions = nullptr; q.rewards = nullptr; q.dones = nullptr; q.rows = nullptr; q.obs_every_step = 0;
        const int nb = pmlp_nb_for(p->policy->hidden), ks = pmlp_ks_for(2 * p->nvars * p->k);
        const size_t lds_pol = (size_t)(threads / WAVE) * binom_scratch_bytes<W>(q.obs_rows);
#define BBX_BPOL(NBV, KSV) hipLaunchKernelGGL((bbx_binom_policy_kernel<W, NBV, KSV>), dim3(blocks), dim3(threads), lds_pol, stream, q, *p->policy)
        if (ks == 6) { if (nb == 2) BBX_BPOL(2, 6); else BBX_BPOL(4, 6); }
        else if (W == 4 && ks == 10) { if (nb == 2) BBX_BPOL(2, 10); else BBX_BPOL(4, 10); }
        else return (int)hipErrorInvalidValue;             // (bbx_api.cpp admits only the built-in shapes)
#undef BBX_BPOL
        return 0;
      }
    }
    lds = lds_plain;
    if (trace) BBX_LAUNCH((bbx_binom_kernel<W, false, true>)); else BBX_LAUNCH((bbx_binom_kernel<W, false, false>));
  return 0;
}
extern "C" int bbx_launch_binom(const BbxParams* p, int kind, int blocks, int threads, size_t lds, hipStream_t stream) {
  return p->L.W == 2 ? launch_binom_w<2>(p, kind, blocks, threads, lds, stream)
       : p->L.W == 4 ? launch_binom_w<4>(p, kind, blocks, threads, lds, stream) : launch_binom_w<8>(p, kind, blocks, threads, lds, stream);
}
#ifdef BBX_PROF_BUILD
extern "C" int bbx_bin_prof_read(unsigned long long* out, int reset) {   // diagnostic build only
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(bbx_bin_prof_acc), 32 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) { unsigned long long z[32] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(bbx_bin_prof_acc), z, sizeof z); }
  return (int)e;
}
#endif
