// Wide class (one workgroup per environment, bbx_wide.h): launcher.  nw = waves per environment.
#include "bbx_device.h"
#include "bbx_wide.h"

extern "C" int bbx_launch_wide(const BbxParams* p, int nw, hipStream_t stream) {
    BbxParams q = *p;                                      // LDS capacities of the workgroup (terms): forced by the caller or
    const int W_ = (int)q.L.W;                             // as large as the residency aimed at allows
    // lean variants: random and external agents let h grow long — reducer tails collect in an LDS accumulator and h is
    // rewritten only when it is full (LAZY); the ordering strategies keep h short — plain eager merges without the
    // accumulator's bookkeeping.  BBX_WIDE_EAGER=1 / =0 force one or the other (experiments).
    const bool strategy = q.agent == BBX_AGENT_DEGREE || q.agent == BBX_AGENT_FIRST || q.agent == BBX_AGENT_NORMAL || q.agent == BBX_AGENT_SUGAR ||
                          q.agent == BBX_AGENT_LAST || q.agent == BBX_AGENT_CODEGREE || q.agent == BBX_AGENT_STRANGE || q.agent == BBX_AGENT_SPICE;
    bool lazy = !q.accounting && !strategy;
    if (const char* ev = getenv("BBX_WIDE_EAGER")) lazy = !q.accounting && ev[0] == '0';
    if (W_ == 8) lazy = false;                             // (32-byte monomials have no sort key: the accumulator holds keys)
    const bool acct = q.accounting != 0;
    bool one_per_cu = false;
    if (q.wide_hc > 0) {                                   // forced capacities (tests, experiments): as asked, as far as 160 KB go
      q.wide_hc = (q.wide_hc + 7) & ~7;
      while (q.wide_hc > 8 && wide_lds_bytes(W_, q.wide_hc, q.wide_hc, q.wide_hc, lazy ? q.wide_hc : 0) > 160u * 1024u) q.wide_hc -= 8;
      q.wide_fc = q.wide_hc; q.wide_rc = q.wide_hc; q.wide_sc = lazy ? q.wide_hc : 0;
    }
    else {
      static int ncu = 0;
      if (!ncu) { int dev = 0; hipDeviceProp_t pr; if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount; if (ncu <= 0) ncu = 256; }
      // one workgroup per CU while the batch fits that way (160 KB each), otherwise two per CU (80 KB each)
      const bool one = q.B <= ncu || q.wide_tail == 2;     // (the tail kernel: the workgroups with work left fit one per CU)
      one_per_cu = one;
      const size_t budget = one ? 160u * 1024u : 80u * 1024u;
      // (without the accumulator's two buffers the reducer table can hold twice as many reducers)
      q.wide_fc = one ? 1024 : 512; q.wide_rc = one ? (lazy ? 1024 : 2048) : (lazy ? 704 : 1024); q.wide_sc = lazy ? (one ? 1536 : 1024) : 0;
      const size_t fixed = wide_lds_bytes(W_, 0, q.wide_fc, q.wide_rc, q.wide_sc);
      q.wide_hc = (int)((budget - fixed) / 20) & ~63;       // a term in LDS: 8-byte sort key + u16 coefficient, two buffers
      while (wide_lds_bytes(W_, q.wide_hc, q.wide_fc, q.wide_rc, q.wide_sc) > budget) q.wide_hc -= 64;
    }
    p = &q;
    const size_t wl = wide_lds_bytes(W_, q.wide_hc, q.wide_fc, q.wide_rc, q.wide_sc);
    const bool tr = p->trace != nullptr;
#define BBX_WIDE_LAUNCH(WW, TT, LL) do { \
      hipError_t err_ = hipFuncSetAttribute((const void*)bbx_wide_kernel<WW, TT, LL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl); \
      if (err_ != hipSuccess) return (int)err_; \
      hipLaunchKernelGGL((bbx_wide_kernel<WW, TT, LL>), dim3(p->B), dim3(nw * WAVE), wl, stream, *p); } while (0)
#define BBX_WIDE_LAUNCH1(WW, LL, AA) do { \
      hipError_t err_ = hipFuncSetAttribute((const void*)bbx_wide_kernel_1cu<WW, LL, AA>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl); \
      if (err_ != hipSuccess) return (int)err_; \
      hipLaunchKernelGGL((bbx_wide_kernel_1cu<WW, LL, AA>), dim3(p->B), dim3(nw * WAVE), wl, stream, *p); } while (0)
#define BBX_WIDE_LAUNCHE(WW) do { \
      hipError_t err_ = hipFuncSetAttribute((const void*)bbx_wide_eager_kernel<WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wl); \
      if (err_ != hipSuccess) return (int)err_; \
      hipLaunchKernelGGL((bbx_wide_eager_kernel<WW>), dim3(p->B), dim3(nw * WAVE), wl, stream, *p); } while (0)
    if (W_ == 8) { if (tr) BBX_WIDE_LAUNCH(8, true, false); else BBX_WIDE_LAUNCH(8, false, false); }   // one variant: tier 3 throughout
    else if (!tr && one_per_cu && !getenv("BBX_WIDE_NO1CU")) {
      if (W_ == 2) { if (lazy) BBX_WIDE_LAUNCH1(2, true, false); else if (acct) BBX_WIDE_LAUNCH1(2, false, true); else BBX_WIDE_LAUNCH1(2, false, false); }
      else { if (lazy) BBX_WIDE_LAUNCH1(4, true, false); else if (acct) BBX_WIDE_LAUNCH1(4, false, true); else BBX_WIDE_LAUNCH1(4, false, false); }
    } else if (!tr && !lazy && !acct) {
      if (W_ == 2) BBX_WIDE_LAUNCHE(2); else BBX_WIDE_LAUNCHE(4);
    } else
    if (W_ == 2) { if (tr) { if (lazy) BBX_WIDE_LAUNCH(2, true, true); else BBX_WIDE_LAUNCH(2, true, false); }
                   else { if (lazy) BBX_WIDE_LAUNCH(2, false, true); else BBX_WIDE_LAUNCH(2, false, false); } }
    else { if (tr) { if (lazy) BBX_WIDE_LAUNCH(4, true, true); else BBX_WIDE_LAUNCH(4, true, false); }
           else { if (lazy) BBX_WIDE_LAUNCH(4, false, true); else BBX_WIDE_LAUNCH(4, false, false); } }
#undef BBX_WIDE_LAUNCH
#undef BBX_WIDE_LAUNCH1
#undef BBX_WIDE_LAUNCHE
    return (int)hipGetLastError();
}
#ifdef BBX_PROF_BUILD
extern "C" int bbx_wide_prof_read(unsigned long long* out, int reset) {   // diagnostic build only
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(bbx_wide_prof_acc), 32 * sizeof(unsigned long long));
  if (e == hipSuccess && reset) { unsigned long long z[32] = {0}; e = hipMemcpyToSymbol(HIP_SYMBOL(bbx_wide_prof_acc), z, sizeof z); }
  return (int)e;
}
#endif
