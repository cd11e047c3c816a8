// Host-side helpers shared by the translation units of libbbx's C ABI (bbx_api.cpp, bbx_alg.cpp).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/bbx.h"
#include "bbx_common.h"

namespace bbx_host {

int fail(int code, const char* fmt, ...);       // records the message bbx_last_error() returns (thread-local); returns code
#define HIPCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bbx_host::fail(BBX_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)

inline uint32_t align16(uint32_t x) { return (x + 15u) & ~15u; }

inline BbxLayout make_layout(int W, int maxG, int maxP, int arena, int maxT) {
  BbxLayout L{};
  L.W = W; L.maxG = maxG; L.maxP = maxP; L.arena = arena; L.maxT = maxT;
  const uint32_t MW = 4u * W;
  uint32_t o = sizeof(BbxHdr);
  auto take = [&o](uint32_t bytes) { uint32_t at = o; o = align16(o + bytes); return at; };
  L.off_lm = take(MW * maxG); L.off_slm = take(MW * maxG); L.off_lcm = take(MW * maxG);
  L.off_am = take(MW * arena); L.off_hm = take(MW * 5u * maxT);
  L.off_poff = take(4u * maxG); L.off_pairs = take(4u * maxP);
  L.off_sidx = take(2u * maxG); L.off_plen = take(2u * maxG); L.off_psug = take(2u * maxG); L.off_pinv = take(2u * maxG);
  L.off_ac = take(2u * arena); L.off_hc = take(2u * 5u * maxT); L.off_cp = take(maxG);
  L.rec_bytes = (o + 255u) & ~255u;
  return L;
}

// binomial class: no arena, fixed two-term polynomials (bbx_common.h)
inline BbxLayout make_layout_binom(int W, int maxG, int maxP) {
  BbxLayout L{};
  L.W = W; L.maxG = maxG; L.maxP = maxP; L.arena = 0; L.maxT = 2; L.kind = 1;
  const uint32_t MW = 4u * W;
  uint32_t o = sizeof(BbxHdr);
  auto take = [&o](uint32_t bytes) { uint32_t at = o; o = align16(o + bytes); return at; };
  L.off_lm = take(MW * maxG); L.off_tm = take(MW * maxG); L.off_slm = take(MW * maxG); L.off_stm = take(MW * maxG);
  L.off_lcm = take(MW * maxG); L.off_ginfo = take(8u * maxG); L.off_sinfo = take(8u * maxG);
  L.off_pairs = take(4u * maxP); L.off_cp = take(maxG);
  L.rec_bytes = (o + 255u) & ~255u;
  return L;
}


// Device and pinned-host buffers of destroyed handles are kept for the next handle (sizes rounded up to powers of two, at
// most 256 MiB of device and 64 MiB of pinned memory cached, blocks of up to 32 MiB): a tree search that calls env.copy()
// per node (mcts.py:89,96,147) creates and destroys one-environment handles by the thousand, and a dozen hipMalloc /
// hipHostMalloc calls per handle were 85 % of such a copy.  Contents are never assumed zero (hipMalloc does not zero either).
hipError_t pool_malloc(void** p, size_t n);
hipError_t pool_free(void* p);
hipError_t pool_host_malloc(void** p, size_t n, unsigned flags);
hipError_t pool_host_free(void* p);
uint16_t* inv_table(int device);                      // GF(32003) inverses on the device (one table per device and process)
void pool_synced(bool on);                          // the calling thread has synchronised the device: frees need not
}  // namespace bbx_host
#ifndef BBX_NO_POOL_MACROS
#define hipMalloc(p, n) bbx_host::pool_malloc((void**)(p), (n))
#define hipFree(p) bbx_host::pool_free((void*)(p))
#define hipHostMalloc(p, n, fl) bbx_host::pool_host_malloc((void**)(p), (n), (fl))
#define hipHostFree(p) bbx_host::pool_host_free((void*)(p))
#endif
