// Buffers of destroyed handles kept for the next handle, and the per-device inverse table (declared in bbx_host.h): the one
// translation unit of libbbx's host side that calls hipMalloc / hipFree / hipHostMalloc / hipHostFree themselves.
#define BBX_NO_POOL_MACROS
#include <hip/hip_runtime.h>
#include <cstdint>
#include <mutex>
#include <unordered_map>
#include <vector>
#include "../../include/bbx.h"
#include "bbx_common.h"
#include "bbx_ideals.h"
#include "bbx_host.h"

namespace bbx_host {

// ---- buffer cache (bbx_host.h) ---------------------------------------------------------------------------------------
namespace {
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, std::pair<size_t, uint64_t>> live;          // pointer -> (rounded size, key)
  std::unordered_map<uint64_t, std::vector<void*>> idle;               // key (device, kind, flags, size class) -> blocks
  size_t cached_dev = 0, cached_pin = 0;
};
Pool& pool() { static Pool* p = new Pool; return *p; }                   // (never destroyed: handles may outlive static destructors)
// Blocks that can be cached (<= POOL_MAX_CACHED) are rounded up to a power of two, their size class; anything larger — record
// arrays of whole batches, gigabytes — is never cached and is allocated as asked for (256-byte granules): rounding those
// up would cost up to twice the memory and make hipMemGetInfo checks of the callers lie.
constexpr size_t POOL_MAX_CACHED = size_t(32) << 20;
size_t pool_round(size_t n) {
  if (n > POOL_MAX_CACHED) return (n + 255) & ~size_t(255);
  size_t r = 256; while (r < n) r <<= 1; return r;
}
// out of memory: give back every idle block of that kind on this device
void pool_release_idle(int dev, int kind) {
  Pool& P = pool();
  std::vector<void*> drop;
  {
    std::lock_guard<std::mutex> g(P.mu);
    for (auto& kv : P.idle) {
      if ((int)(kv.first >> 56) != (dev & 0xff) || (int)((kv.first >> 55) & 1) != (kind & 1)) continue;
      const size_t r = size_t(1) << (kv.first & 0xff);
      for (void* q : kv.second) { drop.push_back(q); (kind ? P.cached_pin : P.cached_dev) -= r; }
      kv.second.clear();
    }
  }
  for (void* q : drop) (void)(kind ? hipHostFree(q) : hipFree(q));
}
uint64_t pool_key(int dev, int kind, unsigned flags, size_t rounded) {
  int cls = 0; while ((size_t(1) << cls) < rounded) cls++;
  return ((uint64_t)(dev & 0xff) << 56) | ((uint64_t)(kind & 1) << 55) | ((uint64_t)(flags & 0xffff) << 32) | (uint64_t)cls;
}
hipError_t pool_get(void** p, size_t n, int kind, unsigned flags) {
  if (n == 0) n = 1;
  int dev = 0; (void)hipGetDevice(&dev);
  const size_t r = pool_round(n);
  const uint64_t key = pool_key(dev, kind, flags, r);
  Pool& P = pool();
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.idle.find(key);
    if (it != P.idle.end() && !it->second.empty()) {
      *p = it->second.back(); it->second.pop_back();
      (kind ? P.cached_pin : P.cached_dev) -= r;
      P.live[*p] = {r, key};
      return hipSuccess;
    }
  }
  hipError_t e = kind ? hipHostMalloc(p, r, flags) : hipMalloc(p, r);
  if (e != hipSuccess) {                                   // the idle cache may be what is in the way: release it and try once more
    (void)hipGetLastError();
    pool_release_idle(dev, kind);
    e = kind ? hipHostMalloc(p, r, flags) : hipMalloc(p, r);
  }
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> g(P.mu);
  P.live[*p] = {r, key};
  return hipSuccess;
}
thread_local bool pool_caller_synced = false;
hipError_t pool_put(void* p, int kind) {
  if (!p) return hipSuccess;
  // (hipFree waits for the device before it releases memory, and callers rely on that: so does this, unless the caller has
  // just synchronised itself — a handle's destructor, which returns a dozen buffers)
  if (!pool_caller_synced) (void)hipDeviceSynchronize();
  Pool& P = pool();
  size_t r = 0; uint64_t key = 0; bool known = false, keep = false;
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.live.find(p);
    if (it != P.live.end()) {
      known = true; r = it->second.first; key = it->second.second; P.live.erase(it);
      size_t& cached = kind ? P.cached_pin : P.cached_dev;
      const size_t cap = kind ? (size_t(64) << 20) : (size_t(256) << 20);
      if (r <= POOL_MAX_CACHED && cached + r <= cap) { P.idle[key].push_back(p); cached += r; keep = true; }
    }
  }
  (void)known;
  if (keep) return hipSuccess;
  return kind ? hipHostFree(p) : hipFree(p);
}
}  // namespace
void pool_synced(bool on) { pool_caller_synced = on; }
hipError_t pool_malloc(void** p, size_t n) { return pool_get(p, n, 0, 0u); }
hipError_t pool_free(void* p) { return pool_put(p, 0); }
hipError_t pool_host_malloc(void** p, size_t n, unsigned flags) { return pool_get(p, n, 1, flags); }
hipError_t pool_host_free(void* p) { return pool_put(p, 1); }
// 1/x mod 32003 for every x, on the device: computed and uploaded once per device and process
uint16_t* inv_table(int device) {
  static std::mutex mu; static uint16_t* tab[64] = {nullptr};
  if (device < 0 || device >= 64) return nullptr;
  std::lock_guard<std::mutex> g(mu);
  if (!tab[device]) {
    std::vector<uint16_t> inv(BBX_P, 0);
    for (uint32_t x = 1; x < BBX_P; x++) inv[x] = (uint16_t)bbx::coef_inv((int)x);
    uint16_t* d = nullptr;
    if (hipMalloc((void**)&d, BBX_P * sizeof(uint16_t)) != hipSuccess) return nullptr;
    if (hipMemcpy(d, inv.data(), BBX_P * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(d); return nullptr; }
    tab[device] = d;
  }
  return tab[device];
}

}  // namespace bbx_host
