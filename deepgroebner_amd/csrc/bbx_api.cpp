// libbbx.so — host side of the C ABI declared in include/bbx.h.
// Owns the device memory (environment records, ideal queues, output buffers), drives the HIP
// kernels in bbx_*.hip and keeps the per-environment ideal generators (bbx_ideals.cpp).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <unordered_map>
#include <random>
#include <string>
#include <vector>
#include <chrono>
#include <atomic>

#include "../../include/bbx.h"
#include "bbx_common.h"
#include "bbx_ideals.h"
#include "bbx_batch.h"

namespace {

thread_local std::string g_err;
}  // namespace
int bbx_host::fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
namespace {
using bbx_host::fail; using bbx_host::make_layout; using bbx_host::make_layout_binom;

struct OutBuf {            // one contiguous device block so a step needs a single D2H copy
  double* rewards; int32_t* rows; uint8_t* dones;
  size_t bytes;
};

}  // namespace

bbx_batch::~bbx_batch() {
  if (!d_recs && !d_q && !d_out && !h_io && !gen_owner) return;   // nothing was ever allocated
  (void)hipSetDevice(device);
  (void)hipDeviceSynchronize();
  bbx_host::pool_synced(true);
  for (auto& ev : ev_open) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  if (d_ctl) (void)hipFree(d_ctl);
  if (d_wide_done) (void)hipFree(d_wide_done);
  if (ps_ev) (void)hipEventDestroy(ps_ev);
  if (ps_stream) (void)hipStreamDestroy(ps_stream);
  if (ps_ctl_stream) (void)hipStreamDestroy(ps_ctl_stream);
  if (d_clone_idx) (void)hipFree(d_clone_idx);
  void* dev[] = {d_recs, d_q, d_tail, d_out, d_actions, d_mask, d_seeds, d_obs, d_trace, d_hdr,
                 d_vrecs, d_vhdr, d_vsrc, d_vseeds, d_vvals, d_stage, d_obs_off, d_obs_packed};
  for (void* q : dev) if (q) (void)hipFree(q);
  for (void* q : retired) if (q) (void)hipFree(q);
  void* pinned[] = {h_io, h_act, h_stage, h_zobs, h_obs, (void*)h_mbox};
  for (void* q : pinned) if (q) (void)hipHostFree(q);
  bbx_host::pool_synced(false);
}

namespace bbx_host {

int pack_mono(const bbx_batch* b, const bbx::HTerm& t, uint32_t* w) {
  const int W = b->W, slots = 2 * W;
  uint32_t s[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int v = 0; v < bbx::kN; v++) {
    if (t.e[v] == 0) continue;
    if (v >= slots - 1) return fail(BBX_E_UNSUPPORTED, "variable index %d does not fit the %d-slot monomial", v, slots);
    if (t.e[v] < 0 || t.e[v] > 65535) return fail(BBX_E_UNSUPPORTED, "exponent %d out of range", t.e[v]);
    s[v] = (uint32_t)t.e[v];
  }
  if (t.deg > 65535) return fail(BBX_E_UNSUPPORTED, "degree %d out of range", t.deg);
  s[slots - 1] = (uint32_t)t.deg;
  for (int i = 0; i < W; i++) w[i] = s[2 * i] | (s[2 * i + 1] << 16);
  return BBX_OK;
}

// serialise one ideal into a queue slot: [npolys, {nterms, sugar, {coef, mono[W]}...}...]
int pack_ideal(const bbx_batch* b, bbx::HIdeal F, uint32_t* slot) {
  if (b->sort_input)   // BuchbergerEnv::reset, buchberger.cpp:301-302 (std::sort, like the reference)
    std::sort(F.begin(), F.end(), [](const bbx::HPoly& f, const bbx::HPoly& g) { return bbx::mono_gt(g.t[0], f.t[0]); });
  size_t need = 1;
  for (auto& f : F) need += 2 + f.t.size() * (1 + b->W);
  if (need > b->slot_words) return fail(BBX_E_CAPACITY, "ideal needs %zu queue words, slot has %u", need, b->slot_words);
  uint32_t* w = slot;
  *w++ = (uint32_t)F.size();
  for (auto& f : F) {
    if (f.t.empty()) return fail(BBX_E_ARG, "zero polynomial among the generators");
    if ((int)f.t.size() > (int)b->L.maxT) return fail(BBX_E_CAPACITY, "generator with %zu terms exceeds max_poly_terms %u", f.t.size(), b->L.maxT);
    *w++ = (uint32_t)f.t.size();
    *w++ = (uint32_t)f.sugar;
    for (auto& t : f.t) {
      *w++ = (uint32_t)t.c;
      int rc = pack_mono(b, t, w);
      if (rc) return rc;
      w += b->W;
    }
  }
  return BBX_OK;
}

int upload_queue(bbx_batch* b, hipStream_t stream = 0) {
  if (!b->q_dirty) return BBX_OK;
  const size_t stride = b->fixed ? b->h_q.size() : (size_t)b->nslots * b->slot_words;
  int nd = 0;
  if (!b->fixed && !b->q_dirty_env.empty()) for (int e = 0; e < b->B; e++) nd += b->q_dirty_env[e] ? 1 : 0;
  if (b->fixed || b->q_dirty_env.empty() || (size_t)nd * 4 > (size_t)b->B) {
    // everything (first fill, prefetch): plain copies
    HIPCHK(hipMemcpy(b->d_q, b->h_q.data(), b->h_q.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(b->d_tail, b->h_tail.data(), b->h_tail.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  } else if (nd > 0) {
    // a few rings (the steady state of host-driven stepping): stage them contiguously in pinned memory, ONE copy, and
    // let a small kernel put them in place — instead of one synchronous copy per run of refilled environments
    const size_t need = 2 * (size_t)nd + (size_t)nd * stride;
    if (need > b->stage_words) {
      if (b->h_stage) (void)hipHostFree(b->h_stage);
      if (b->d_stage) (void)hipFree(b->d_stage);
      b->h_stage = nullptr; b->d_stage = nullptr; b->stage_words = 0;
      const size_t cap = need * 2;
      HIPCHK(hipHostMalloc((void**)&b->h_stage, cap * sizeof(uint32_t), hipHostMallocDefault));
      HIPCHK(hipMalloc((void**)&b->d_stage, cap * sizeof(uint32_t)));
      b->stage_words = cap;
    }
    int i = 0;
    for (int e = 0; e < b->B; e++) {
      if (!b->q_dirty_env[e]) continue;
      b->h_stage[i] = (uint32_t)e;
      b->h_stage[nd + i] = (uint32_t)b->h_tail[e];
      memcpy(b->h_stage + 2 * (size_t)nd + (size_t)i * stride, b->h_q.data() + (size_t)e * stride, stride * sizeof(uint32_t));
      i++;
    }
    HIPCHK(hipMemcpy(b->d_stage, b->h_stage, need * sizeof(uint32_t), hipMemcpyHostToDevice));   // pinned: returns once copied
    int lrc = bbx_launch_scatter_queue(b->d_stage, nd, (uint32_t)stride, b->d_q, b->d_tail, stream);
    if (lrc) return fail(BBX_E_DEVICE, "queue scatter launch failed: %s", hipGetErrorString((hipError_t)lrc));
  }
  if (!b->q_dirty_env.empty()) std::fill(b->q_dirty_env.begin(), b->q_dirty_env.end(), 0);
  b->q_dirty = false;
  return BBX_OK;
}

// refill the ring of every environment that holds fewer than min_avail pre-generated ideals
// (launches pass 1: only rings that are empty; bbx_prefetch passes the ring size: top everything up)
int fill_queues(bbx_batch* b, int min_avail, hipStream_t stream) {
  if (b->device_gen) return BBX_OK;                 // the kernels draw their own ideals
  if (b->fixed) return upload_queue(b, stream);
  std::string err;
  bbx::HIdeal F;
  if (b->q_dirty_env.size() != (size_t)b->B) b->q_dirty_env.assign(b->B, b->q_dirty ? 1 : 0);
  for (int e = 0; e < b->B; e++) {
    if (b->h_tail[e] - b->h_head[e] >= std::min(min_avail, (int)b->nslots)) continue;
    while (b->h_tail[e] - b->h_head[e] < (int)b->nslots) {
      // A generator failure (the reference throws: e.g. no two distinct monomials after 1000 trials) is reported when
      // the environment NEEDS that ideal, as in the reference, not when it is drawn ahead of time: the failure is
      // parked and the ring not topped up any further; the draws it consumed stay consumed, exactly as after a caught
      // exception.
      if (!b->gen_error.empty() && !b->gen_error[e].empty()) break;   // parked: finish() raises it if the environment starves
      if (!b->gens[e]->next(F, &err)) {
        if (b->gen_error.size() != (size_t)b->B) b->gen_error.assign(b->B, std::string());
        b->gen_error[e] = err;
        continue;
      }
      uint32_t* slot = b->h_q.data() + (size_t)e * b->nslots * b->slot_words + (size_t)(b->h_tail[e] % (int)b->nslots) * b->slot_words;
      int rc = pack_ideal(b, F, slot);
      if (rc) return rc;
      b->h_tail[e]++;
      b->q_dirty = true;
      b->q_dirty_env[e] = 1;
    }
  }
  return upload_queue(b, stream);
}

int read_headers(bbx_batch* b, hipStream_t stream) {
  b->h_hdr.resize(b->B);
  int lrc = bbx_launch_gather_hdr(b->d_recs, b->L.rec_bytes, b->B, b->d_hdr, stream);
  if (lrc) return fail(BBX_E_DEVICE, "gather launch failed: %s", hipGetErrorString((hipError_t)lrc));
  HIPCHK(hipMemcpyAsync(b->h_hdr.data(), b->d_hdr, (size_t)b->B * sizeof(BbxHdr), hipMemcpyDeviceToHost, stream));
  HIPCHK(hipStreamSynchronize(stream));
  for (int e = 0; e < b->B; e++) b->h_head[e] = b->h_hdr[e].q_head;
  return BBX_OK;
}

const char* status_name(int s) {
  switch (s) {
    case BBX_ST_G_FULL: return "basis capacity (max_basis) exceeded";
    case BBX_ST_P_FULL: return "pair capacity (max_pairs) exceeded";
    case BBX_ST_ARENA_FULL: return "term arena (arena_terms) exhausted";
    case BBX_ST_POLY_TOO_LONG: return "intermediate polynomial longer than max_poly_terms";
    case BBX_ST_DEG_OVERFLOW: return "degree above 65535";
    case BBX_ST_BAD_ACTION: return "action index outside [0, rows)";
    case BBX_ST_RUNAWAY: return "reduction did not terminate within 2^24 rounds";
    case BBX_ST_POLY_LIMIT: return "a basis element with more than 65535 terms";
    default: return "unknown";
  }
}

void fill_params(bbx_batch* b, BbxParams* p) {
  memset(p, 0, sizeof *p);
  p->recs = b->d_recs; p->L = b->L; p->LL = b->LL; p->B = b->B; p->fast_G = b->fast_G; p->fast_P = b->fast_P;
  p->q.words = b->d_q; p->q.env_stride = b->fixed ? 0 : b->nslots * b->slot_words; p->q.slot_words = b->slot_words;
  p->q.nslots = b->nslots; p->q.fixed = b->fixed ? 1 : 0; p->q.no_redraw = b->listed ? 1 : 0; p->q.tail = b->d_tail;
  p->elim = b->elim; p->rewards_mode = b->rewards; p->sort_reducers = b->sort_reducers; p->k = b->k; p->nvars = b->nvars;
  p->sort_input = (b->device_gen && b->sort_input) ? 1 : 0;
  p->trace = b->d_trace; p->trace_stride = b->trace_cap;
  p->inv_table = b->d_inv;
  p->accounting = b->accounting ? 1 : 0;
  p->lite = b->d_lite;
  p->gen = b->device_gen ? b->d_gen : nullptr;
  p->wide_hc = b->wide_terms;
  p->ctl_stats = b->d_ctl ? b->d_ctl + 8 : nullptr;
}

// enqueue the kernels of one logical launch: the LDS-staged pass (when the class allows) followed by the
// HBM-resident pass that serves whatever the first could not hold; aux launches (nsteps == 0) use one kernel
int enqueue(bbx_batch* b, const BbxParams& p0, bool resume, hipStream_t stream) {
  BbxParams p = p0;
  int kinds[3]; int nk = 0;
  if (p.nsteps == 0 && !resume) kinds[nk++] = 2;
  else if (p.ctl) { kinds[nk++] = 3; kinds[nk++] = 0; }   // a kernel of a persistent session, and behind it the HBM-resident
                                                      // class for the environments that outgrew the register/LDS class
  else if (b->wide) kinds[nk++] = 4;
  else if (b->gen_to_wide) {                          // general class, <= 7 variables: wave-per-environment kernel, and behind it the
    kinds[nk++] = 0; kinds[nk++] = 4;                 // workgroup-per-environment kernel for the environments whose polynomials got long
    p.spill_terms = 384;
  }
  else {   // the register/LDS-resident class (the hand-tuned kernel where the batch has it), then the HBM-resident one
    const bool pol_hbm_only = p.policy && p.policy->rollout == 2;     // a policy rollout outside the register/LDS class
    if (b->staged && !pol_hbm_only) kinds[nk++] = b->fast ? 3 : 1;   // (the hand-tuned kernel knows every agent since round 4)
    // the HBM-resident pass behind the LDS-resident one serves environments that outgrow the LDS class inside a
    // rollout; a single host-driven step does without it: an environment that spills reports BBX_ST_SPILL and
    // finish() continues it (one launch less on the latency path)
    // (asynchronous calls on caller buffers always get it: nobody polls their status words between steps)
    if (!b->staged || pol_hbm_only || resume || p.nsteps > 1 || b->obs_external || b->device_async) kinds[nk++] = 0;
  }
  // wide class with more workgroups than CUs: a second kernel for the tail of the launch (BbxParams::wide_tail)
  int tail_at = -1;
  if (nk > 0 && kinds[nk - 1] == 4 && nk < 3 && b->d_wide_done && b->B > b->ncu && p.L.W <= 4 && !getenv("BBX_NO_WIDE_TAIL")) {
    HIPCHK(hipMemsetAsync(b->d_wide_done, 0, 256, stream));
    tail_at = nk; kinds[nk++] = 4;
    p.wide_done = b->d_wide_done; p.wide_ncu = b->ncu;
  }
  // a host-driven zero-copy step whose only kernel is the hand-tuned one: the host spins on the status words in pinned
  // memory instead of waiting for the runtime's completion signal (read_lite)
  b->poll_active = !resume && nk == 1 && kinds[0] == 3 && b->zc_active && p.lite != nullptr && !b->timing && b->poll_misses < 3 && !getenv("BBX_NO_POLL");
  if (b->poll_active) {
    b->poll_seq = (b->poll_seq % 16000) + 1; p.done_seq = b->poll_seq;
    // the words the host is going to watch start out cleared: pinned memory is handed out uninitialised and may still hold
    // the status words — sequence numbers included — of a handle that was destroyed
    for (int e = 0; e < b->B; e++) ((volatile int32_t*)b->h_io)[(size_t)e * 4] = 0;
    std::atomic_thread_fence(std::memory_order_release);
  } else p.done_seq = 0;
  for (int i = 0; i < nk; i++) {
    if (resume || i > 0) { p.set_budget = 0; p.pass = 1; }
    p.wide_tail = tail_at < 0 ? 0 : (i == tail_at ? 2 : (i == tail_at - 1 ? 1 : 0));
    if (i > 0 && p0.ctl) { p.ctl = nullptr; p.sess_target = p0.nsteps; }   // (what is owed of the session's total when it runs; slice_ticks
                                                                           // != 0 tells it that the host looks after environments it hands back)
    // a per-step policy call: only the first pass of the fast class evaluates the policy (the follow-up reads its actions);
    // a policy rollout: the HBM-resident continuation pass has the policy too
    if (p.policy && !(p.policy->rollout ? (!resume && (kinds[i] == 3 || kinds[i] == 0)) : (!resume && i == 0 && kinds[i] == 3))) p.policy = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    // the primary (dominant) kernel of the sequence; where long polynomials continue in the wide kernel, that one too
    const bool timed = b->timing && kinds[i] != 2 && (i == 0 || kinds[i] == 4);
    if (timed) {
      HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
      HIPCHK(hipEventRecord(e0, stream));
    }
    int lrc = bbx_launch_step(&p, kinds[i], kinds[i] == 4 ? (b->wide ? b->wide : 8) : b->envs_per_block, stream);
    if (lrc) return fail(BBX_E_DEVICE, "kernel launch failed: %s", hipGetErrorString((hipError_t)lrc));
    b->step_kernels++;
    if (timed) { HIPCHK(hipEventRecord(e1, stream)); b->ev_open.push_back({e0, e1}); }
  }
  return BBX_OK;
}

int collect_events(bbx_batch* b) {
  for (auto& ev : b->ev_open) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ev.first, ev.second));
    b->kernel_ms += ms; b->kernel_launches++;
    (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second);
  }
  b->ev_open.clear();
  return BBX_OK;
}

// engine state of std::default_random_engine after seed(s) (MinStd0::seed), written into the record headers
int write_gen_states(bbx_batch* b, const std::vector<long long>& seeds) {
  std::vector<uint32_t> st(b->B);
  for (int e = 0; e < b->B; e++) { bbx::MinStd0 r; r.seed(seeds[e]); st[e] = (uint32_t)r.x; }
  HIPCHK(hipMemcpy2D(b->d_recs + offsetof(BbxHdr, gen_rng), b->L.rec_bytes, st.data(), sizeof(uint32_t), sizeof(uint32_t), b->B, hipMemcpyHostToDevice));
  return BBX_OK;
}

int alloc_io(bbx_batch* b, int batch) {
  b->io_bytes = (size_t)batch * 29;
  HIPCHK(hipMalloc((void**)&b->d_out, b->io_bytes));
  HIPCHK(hipMemset(b->d_out, 0, b->io_bytes));
  b->d_lite = (int32_t*)b->d_out;
  b->d_rewards = (double*)(b->d_out + (size_t)batch * 16);
  b->d_rows = (int32_t*)(b->d_out + (size_t)batch * 24);
  b->d_dones = (uint8_t*)(b->d_out + (size_t)batch * 28);
  // (the block the step kernels write their outputs and status words to in zero-copy launches, and the host may spin on:
  // fine-grained, so that device writes are visible while the kernel is still running)
  if (hipHostMalloc((void**)&b->h_io, b->io_bytes, hipHostMallocCoherent | hipHostMallocMapped) != hipSuccess) {
    (void)hipGetLastError();
    HIPCHK(hipHostMalloc((void**)&b->h_io, b->io_bytes, hipHostMallocDefault));
  }
  memset(b->h_io, 0, b->io_bytes);
  HIPCHK(hipHostMalloc((void**)&b->h_act, (size_t)batch * sizeof(int32_t), hipHostMallocDefault));
  memset(b->h_io, 0, b->io_bytes);
  // (host-stepped batches of up to 64 environments: B = 16 / 32 / 64 step in 24 / 28 / 32 us this way, 49 / 59 / 63 us with device
  // buffers and copy calls — scripts/exp_small_batch.py)
  { const char* zm = getenv("BBX_ZERO_COPY_MAX"); b->zero_copy = batch <= (zm ? atoi(zm) : 64) && !getenv("BBX_NO_ZERO_COPY"); }
  if (b->zero_copy) {
    HIPCHK(hipHostGetDevicePointer((void**)&b->zc_io_dev, b->h_io, 0));
    HIPCHK(hipHostGetDevicePointer((void**)&b->zc_act_dev, b->h_act, 0));
  }
  return BBX_OK;
}

// fetch the block the kernels of the last launch left behind (status words and the host-API outputs) in one copy
int read_lite(bbx_batch* b, hipStream_t stream) {
  b->h_lite.resize((size_t)b->B * 4);
  if (!b->zc_active) HIPCHK(hipMemcpyAsync(b->h_io, b->d_out, b->io_bytes, hipMemcpyDeviceToHost, stream));
  bool seen = false;
  if (b->zc_active && b->poll_active) {              // spin on the status words the kernel writes last (a few microseconds
    const volatile int32_t* w = (const volatile int32_t*)b->h_io;   // earlier than the runtime's signal); 2 ms, then the normal wait
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; spins++) {
      bool all = true;
      for (int e = 0; e < b->B; e++) all = all && (((uint32_t)w[(size_t)e * 4]) >> 17) == (uint32_t)b->poll_seq;
      if (all) { seen = true; b->poll_misses = 0; break; }
      if ((spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    if (!seen) b->poll_misses++;                  // (three in a row: the writes do not arrive early on this system; stop spinning)
    b->poll_active = false;
  }
  // (every so often the runtime gets its wait as well, so that it can retire the commands it queued)
  if (!seen || (++b->polled_launches & 63) == 0) HIPCHK(hipStreamSynchronize(stream));   // (zero-copy launches wrote h_io themselves)
  memcpy(b->h_lite.data(), b->h_io, (size_t)b->B * 16);
  for (int e = 0; e < b->B; e++) b->h_head[e] = b->h_lite[(size_t)e * 4 + 1];
  return BBX_OK;
}

// Capacity is a performance cliff, not a failure (the reference's polynomials are heap vectors, polynomials.h:71-94, and
// its basis and pair set grow without bound, buchberger.cpp:52-99): an environment whose next step does not fit its record
// stops BEFORE that step (the kernels check every capacity ahead of the first persistent write, bbx_common.h) and reports
// which array was full; here every such array doubles, all records move to the new layout (one kernel, live prefixes only)
// and the launch is resumed.  `need`: bit s set = some environment reported status s.
int grow_records(bbx_batch* b, unsigned need, int env, hipStream_t stream) {
  uint32_t maxG = b->L.maxG, maxP = b->L.maxP, arena = b->L.arena, maxT = b->L.maxT;
  const char* what = "";
  if (need & (1u << BBX_ST_G_FULL)) {
    what = status_name(BBX_ST_G_FULL);
    if (maxG >= 65534u) return fail(BBX_E_CAPACITY, "environment %d: %s, and a basis cannot exceed 65534 elements (16-bit pair indices)", env, what);
    maxG = std::min(65534u, maxG * 2u);
  }
  if (need & (1u << BBX_ST_P_FULL)) {
    what = status_name(BBX_ST_P_FULL);
    if (maxP >= (1u << 28)) return fail(BBX_E_CAPACITY, "environment %d: %s beyond 2^28 pairs", env, what);
    maxP *= 2u;
  }
  if (need & (1u << BBX_ST_POLY_TOO_LONG)) {
    what = status_name(BBX_ST_POLY_TOO_LONG);
    if (b->binom || maxT >= (1u << 22)) return fail(BBX_E_CAPACITY, "environment %d: %s beyond 2^22 terms", env, what);
    maxT *= 2u;
  }
  if (need & (1u << BBX_ST_ARENA_FULL)) { what = status_name(BBX_ST_ARENA_FULL); arena *= 2u; }
  if (!b->binom) while (arena < 4u * maxT) arena *= 2u;     // every step wants room for one more element of up to maxT terms
  if (!b->binom && arena > (1u << 27)) return fail(BBX_E_CAPACITY, "environment %d: %s beyond 2^27 terms", env, what);
  // pairs of a Gebauer-Moeller step: at most |G| new ones
  while (maxP < 2u * maxG) maxP *= 2u;
  {                                                         // (the layout's offsets are 32-bit: size it in 64 bits first)
    const uint64_t MW = 4ull * b->W;
    const uint64_t est = b->binom ? 128ull + (5ull * MW + 17ull) * maxG + 4ull * maxP
                                  : 128ull + (3ull * MW + 13ull) * maxG + 4ull * maxP + (MW + 2ull) * ((uint64_t)arena + 5ull * maxT);
    if (est > 0xE0000000ull) return fail(BBX_E_CAPACITY, "environment %d: %s, and a record cannot exceed 3.5 GiB", env, what);
  }
  const BbxLayout NL = b->binom ? make_layout_binom(b->W, (int)maxG, (int)maxP) : make_layout(b->W, (int)maxG, (int)maxP, (int)arena, (int)maxT);
  const size_t bytes = (size_t)b->B * NL.rec_bytes;
  size_t freeb = 0, totalb = 0;
  HIPCHK(hipStreamSynchronize(stream));
  if (b->d_vrecs) {                                         // value() scratch is sized by the old layout: rebuilt on demand
    void* old[] = {b->d_vrecs, b->d_vhdr, b->d_vsrc, b->d_vseeds, b->d_vvals};
    for (void* q : old) (void)hipFree(q);
    b->d_vrecs = nullptr; b->d_vhdr = nullptr; b->d_vsrc = nullptr; b->d_vseeds = nullptr; b->d_vvals = nullptr; b->vcap = 0;
  }
  HIPCHK(hipMemGetInfo(&freeb, &totalb));
  if (bytes + (256u << 20) > freeb)
    return fail(BBX_E_CAPACITY, "environment %d: %s, and the device has no room for larger records (%zu MiB needed, %zu MiB free)",
                env, what, bytes >> 20, freeb >> 20);
  char* nrecs = nullptr;
  HIPCHK(hipMalloc((void**)&nrecs, bytes));
  int lrc = bbx_launch_relayout(b->d_recs, nrecs, &b->L, &NL, b->B, stream);
  if (lrc) { (void)hipFree(nrecs); return fail(BBX_E_DEVICE, "relayout launch failed: %s", hipGetErrorString((hipError_t)lrc)); }
  HIPCHK(hipStreamSynchronize(stream));
  if (b->cap_valid) {
    // a recorded graph (bbx_graph_replayed) holds the old array's address: replays of it must stay harmless — they step the
    // retired copy, not memory that has meanwhile been handed to someone else — until the caller hears about it and records again
    b->retired.push_back(b->d_recs);
    b->cap_stale = true;
  } else HIPCHK(hipFree(b->d_recs));
  b->d_recs = nrecs; b->L = NL;
  b->last.recs = nrecs; b->last.L = NL;
  b->grow_events++;
  if (getenv("BBX_VERBOSE"))
    fprintf(stderr, "[bbx] records enlarged (%s, environment %d): max_basis %u max_pairs %u arena_terms %u max_poly_terms %u, %zu MiB\n",
            what, env, maxG, maxP, arena, maxT, bytes >> 20);
  return BBX_OK;
}

// wait for the launch in flight; serve environments that ran out of queued ideals or outgrew the LDS class; surface
// errors.  Whatever happens, the handle is left with nothing in flight: an error is reported once, not re-raised by
// every later call, and environments that only needed service (STARVED / SPILL) have been served before the first
// error of another environment is returned.
int finish_impl(bbx_batch* b, hipStream_t stream);

int finish(bbx_batch* b, hipStream_t stream) {
  b->api_epoch++;
  const bool mbox = b->ps_active && b->ps_mbox;              // (closing a mailbox session switches the pinned outputs on: off again behind it)
  if (mbox) {
    // nothing owed (the host waited for every step it issued): tell the waves to go, wait for the kernel, done — no closing kernel
    bool settled = true;
    const volatile int32_t* w = (const volatile int32_t*)b->h_io;
    const uint32_t want = (uint32_t)(b->ps_target % 16000) + 1u;
    for (int e = 0; e < b->B; e++) { const uint32_t v = (uint32_t)w[(size_t)e * 4]; settled = settled && (v >> 17) == want && (v & 0xffffu) == BBX_ST_OK; }
    if (settled) {
      b->ps_active = false;
      int rc0 = ps_write_ctl(b, true);
      if (rc0) return rc0;
      b->ps_mbox = false;
      // every wave marks its status block when it has stored its environment and left (bbx_fast.h): a spin of a few
      // microseconds instead of the runtime's wait for the kernel (hundreds, once per episode of a gym loop)
      bool gone = false;
      const auto t0 = std::chrono::steady_clock::now();
      for (unsigned spins = 0; !gone; spins++) {
        gone = true;
        for (int e = 0; e < b->B; e++) gone = gone && (w[(size_t)e * 4 + 2] & 0x40000000);
        if (!gone && (spins & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
      }
      std::atomic_thread_fence(std::memory_order_acquire);
      if (!gone) HIPCHK(hipStreamSynchronize(b->ps_stream));
      b->last = b->ps_p; b->last.recs = b->d_recs; b->last.L = b->L; b->last.ctl = nullptr; b->last.sess_target = 0; b->last.set_budget = 0;
      b->last.policy = nullptr; b->last.mbox = 0;
      b->last_stream = b->ps_stream;
      b->h_lite.resize((size_t)b->B * 4);
      memcpy(b->h_lite.data(), b->h_io, (size_t)b->B * 16);
      for (int e = 0; e < b->B; e++) { b->h_lite[(size_t)e * 4 + 2] &= 0x3fffffff; b->h_head[e] = b->h_lite[(size_t)e * 4 + 1]; }
      b->in_flight = false; b->async_chain = 0;
      return BBX_OK;                                        // (every status word said OK: nothing to serve, nothing to report)
    }
  }
  const int rc = finish_impl(b, stream);
  if (mbox) b->zc_active = false;
  return rc;
}

int finish_impl(bbx_batch* b, hipStream_t stream) {
  int err = BBX_OK;
  const int async_chain = b->async_chain;              // (asynchronous external-action steps behind this wait)
  b->async_chain = 0;
  auto note = [&err](int code) { if (err == BBX_OK) err = code; };
  if (b->ps_active) {                                // a persistent session: stop it; its kernels run in slices until nothing is owed
    int rc = session_close(b, false, nullptr, true);
    if (rc) { b->in_flight = false; return rc; }
    stream = b->ps_stream;
    for (int guard = 0;; guard++) {
      rc = read_lite(b, stream);
      if (rc) { b->in_flight = false; return rc; }
      bool owed = false;
      for (int e = 0; e < b->B && !owed; e++) owed = (b->h_lite[(size_t)e * 4] & 0xffff) == BBX_ST_TIMESLICE;
      if (!owed) break;
      if (guard >= 100000) { b->in_flight = false; return fail(BBX_E_DEVICE, "a persistent session still owes steps after 100000 time slices"); }
      rc = session_kernel(b, false, nullptr, true);
      if (rc) { b->in_flight = false; return rc; }
    }
  }
  for (int round = 0;; round++) {
    int rc = read_lite(b, stream);
    if (rc) { b->in_flight = false; return rc; }
    rc = collect_events(b);
    if (rc) { b->in_flight = false; return rc; }
    bool again = false;
    unsigned grow = 0; int grow_env = -1;
    for (int e = 0; e < b->B; e++) {
      const int st = b->h_lite[(size_t)e * 4] & 0xffff;
      if (st == BBX_ST_STARVED && !b->gen_error.empty() && !b->gen_error[e].empty() && b->h_tail[e] - b->h_head[e] <= 0) {
        if (err == BBX_OK) {
          const std::string msg = b->gen_error[e];       // the draw this environment is waiting for is the one that failed
          b->gen_error[e].clear();
          note(fail(BBX_E_GENERATOR, "%s", msg.c_str()));
        }
        continue;
      }
      if (st == BBX_ST_GEN_ZERO) { if (err == BBX_OK) note(fail(BBX_E_GENERATOR, "random polynomial cancelled to zero (undefined in the reference)")); }
      else if (st == BBX_ST_GEN_FAIL) { if (err == BBX_OK) note(fail(BBX_E_GENERATOR, "failed to generate two distinct random monomials after 1000 trials")); }
      else if ((st == BBX_ST_STARVED || st == BBX_ST_SPILL) && b->policy_rollout) {
        // (the continuation pass runs right behind the first one; what is still unfinished here cannot be resumed: the
        // policy arguments belonged to the caller's frame)
        if (err == BBX_OK) note(fail(BBX_E_CAPACITY, "environment %d could not finish its policy rollout (%s)", e, status_name(st)));
      }
      else if (st == BBX_ST_STARVED || st == BBX_ST_SPILL) again = true;
      else if (st == BBX_ST_BAD_ACTION) { if (err == BBX_OK) note(fail(BBX_E_ACTION, "environment %d: %s", e, status_name(st))); }
      else if (bbx_st_capacity(st) && !b->no_growth) { grow |= 1u << st; if (grow_env < 0) grow_env = e; }
      else if (st != BBX_ST_OK && err == BBX_OK) {
        rc = read_headers(b, stream);
        if (rc) { b->in_flight = false; return rc; }
        note(fail(BBX_E_CAPACITY, "environment %d: %s (|G|=%d |P|=%d terms=%d)", e, status_name(st),
                  b->h_hdr[e].nG, b->h_hdr[e].nP, b->h_hdr[e].arena_used));
      }
    }
    if (grow) {                                       // enlarge what was full; the environments then take the step they stopped at
      rc = grow_records(b, grow, grow_env, stream);
      if (rc) { note(rc); break; }
      if (b->policy_rollout) {                        // (its per-step arrays belonged to the caller's frame: cannot be resumed)
        note(fail(BBX_E_CAPACITY, "environment %d could not finish its policy rollout (%s); the records have been enlarged, later rollouts have room",
                  grow_env, status_name(__builtin_ctz(grow))));
        break;
      }
      if (b->device_async && b->last.agent == BBX_AGENT_EXTERNAL && async_chain > 1) {
        // several asynchronous steps with caller-supplied actions were queued behind each other (or replayed from a graph):
        // the environment stopped at one of them and sat out the rest; the action buffer now holds a later step's actions,
        // so the step it stopped at cannot be taken for it
        note(fail(BBX_E_CAPACITY, "environment %d outgrew its records (%s) inside a chain of asynchronous steps with caller-supplied actions and took "
                                  "none of the chain's later steps; the records have been enlarged, later calls have room", grow_env,
                  status_name(__builtin_ctz(grow))));
        break;
      }
      again = true;
    }
    if (!again) break;
    if (round > 100000) { note(fail(BBX_E_GENERATOR, "ideal queue starvation did not resolve")); break; }
    rc = fill_queues(b, 1, stream);
    if (rc) { b->in_flight = false; return rc; }
    rc = enqueue(b, b->last, true, stream);   // continue the rollout where each environment stopped
    if (rc) { b->in_flight = false; return rc; }
  }
  b->in_flight = false;
  if (err == BBX_OK && b->obs_external)
    for (int e = 0; e < b->B; e++)
      if (b->h_lite[(size_t)e * 4] & BBX_LITE_OBS_TRUNC)
        return fail(BBX_E_CAPACITY, "environment %d: an observation had more rows than the caller's block holds (obs_rows = %d) or, in a policy "
                                    "rollout, than the policy kernels score (%d); the extra rows were not written / scored", e, b->last.obs_rows, BBX_POLICY_MAX_ROWS);
  return err;
}

// zero-copy launches: outputs and status words go straight to the pinned host block
void zc_outputs(bbx_batch* b, BbxParams* p) {
  p->lite = (int32_t*)b->zc_io_dev;
  p->rewards = (double*)(b->zc_io_dev + (size_t)b->B * 16);
  p->rows = (int32_t*)(b->zc_io_dev + (size_t)b->B * 24);
  p->dones = (uint8_t*)(b->zc_io_dev + (size_t)b->B * 28);
}

// the outputs of a host-API launch: already on the host (finish() fetched the whole block)
int copy_out(bbx_batch* b, double* rewards, uint8_t* dones, int32_t* rows) {
  if (rewards) memcpy(rewards, b->h_io + (size_t)b->B * 16, (size_t)b->B * 8);
  if (rows) memcpy(rows, b->h_io + (size_t)b->B * 24, (size_t)b->B * 4);
  if (dones) memcpy(dones, b->h_io + (size_t)b->B * 28, (size_t)b->B);
  return BBX_OK;
}

int create_common(std::unique_ptr<bbx::IdealGen> proto, int nvars_obs, int elimination, int rewards, int sort_input,
                  int sort_reducers, int k, int batch, int device, const bbx_caps* caps, bbx_batch** out,
                  const std::shared_ptr<const std::vector<bbx::HIdeal>>& list = nullptr) {
  if (!out) return fail(BBX_E_ARG, "out is null");
  *out = nullptr;
  if (batch < 1 || k < 1) return fail(BBX_E_ARG, "batch and k must be positive");
  if (elimination < 0 || elimination > 2 || rewards < 0 || rewards > 1) return fail(BBX_E_ARG, "bad elimination/rewards selector");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(BBX_E_DEVICE, "no HIP device available (libbbx has no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(BBX_E_DEVICE, "device %d out of range (have %d)", device, ndev);
  HIPCHK(hipSetDevice(device));

  auto b = std::make_unique<bbx_batch>();
  b->B = batch; b->device = device; b->k = k;
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess) b->ncu = cus; }
  b->elim = elimination; b->rewards = rewards; b->sort_input = sort_input ? 1 : 0; b->sort_reducers = sort_reducers ? 1 : 0;
  b->fixed = proto->fixed();
  b->listed = list != nullptr;
  b->nvars = nvars_obs > 0 ? nvars_obs : proto->nvars();
  // ring variables actually used: probe one ideal from a clone (does not disturb the prototype's stream)
  int maxvar = 0;
  {
    auto probe = proto->clone();
    bbx::HIdeal F; std::string err;
    if (!probe->next(F, &err)) return fail(BBX_E_GENERATOR, "%s", err.c_str());
    for (auto& f : F) for (auto& t : f.t) for (int v = 0; v < bbx::kN; v++) if (t.e[v]) maxvar = std::max(maxvar, v + 1);
    if (list) for (auto& I : *list) for (auto& f : I) for (auto& t : f.t) for (int v = 0; v < bbx::kN; v++) if (t.e[v]) maxvar = std::max(maxvar, v + 1);
    if (!b->fixed && !list) maxvar = std::max(maxvar, proto->nvars());
  }
  if (maxvar > bbx::kN || b->nvars > bbx::kN) return fail(BBX_E_UNSUPPORTED, "more than %d variables", bbx::kN);
  // words per packed monomial: 8 bytes (<= 3 variables), 16 bytes (<= 7), 32 bytes (the reference's N = 8, polynomials.h:29)
  b->W = maxvar <= 3 ? 2 : (maxvar <= 7 ? 4 : 8);
  const bool binomial = !b->fixed && !list && proto->max_terms_hint() == 2;
  bbx_caps c{};
  if (caps) c = *caps;
  if (b->fixed || list) {
    if (!c.max_basis) c.max_basis = 4096; if (!c.max_pairs) c.max_pairs = 16384;
    if (!c.max_poly_terms) c.max_poly_terms = 65536;
    if (!c.arena_terms) {   // long polynomials (cyclic-7: ~2000 terms per element after a few hundred steps): as much
                            // arena as a 48 GiB budget for the whole batch allows, between 2^18 and 2^22 terms
      const long long per_env = (48LL << 30) / batch / 18;
      c.arena_terms = (int)std::max(1LL << 18, std::min(1LL << 22, per_env));
    }
  } else if (binomial) {
    if (!c.max_basis) c.max_basis = b->W == 2 ? 512 : 4096; if (!c.max_pairs) c.max_pairs = b->W == 2 ? (elimination == BBX_GEBAUERMOELLER ? 4096 : 32768) : 16384;
    if (!c.arena_terms) c.arena_terms = 2 * c.max_basis + 16; if (!c.max_poly_terms) c.max_poly_terms = 8;
  } else {
    if (!c.max_basis) c.max_basis = 2048; if (!c.max_pairs) c.max_pairs = 8192;
    if (!c.arena_terms) c.arena_terms = 1 << 18; if (!c.max_poly_terms) c.max_poly_terms = 4096;
  }
  if (!c.queue_slots) c.queue_slots = 8;
  // every array of a record starts 16-byte aligned; the hand-tuned kernel derives the array offsets of the 8-byte
  // monomial layout from the basis capacity alone (bbx_fast.h F_HBM_PTRS), which is exact for even capacities
  if (c.max_basis & 1) c.max_basis += c.max_basis < 65535 ? 1 : -1;
  b->binom = binomial && !c.general_class;
  // long-polynomial environments (fixed ideals such as cyclic-n) in small batches: one workgroup per environment
  if (b->fixed || list) b->wide = c.wide_waves > 0 ? std::min(8, c.wide_waves) : (c.wide_waves < 0 ? 0 : (batch <= 4096 ? 8 : 0));
  if (c.wide_lds_terms < 0 || c.wide_lds_terms > 4096) return fail(BBX_E_ARG, "wide_lds_terms out of range");
  b->wide_terms = c.wide_lds_terms;
  b->no_growth = c.no_growth != 0;
  // LDS-resident class: small binomial environments work out of LDS for the whole launch; anything that
  // outgrows it continues in the HBM-resident pass of the same launch sequence
  b->staged = 0;
  if (binomial && b->W == 2 && c.lds_max_basis >= 0) {
    int lg = c.lds_max_basis ? c.lds_max_basis : 128;
    lg = std::min((lg + 15) & ~15, c.max_basis);       // the working copy never exceeds the HBM record
    b->LL = b->binom ? make_layout_binom(b->W, lg, std::min(2 * lg, c.max_pairs))
                     : make_layout(b->W, lg, std::min(2 * lg, c.max_pairs), std::min(2 * lg + 16, c.arena_terms), c.max_poly_terms);
    b->staged = 1;
    // the hand-tuned kernel covers exactly the reference C++ class's fixed options; its registers and LDS hold bases of
    // 256 elements (bbx_fast.h FLay), independently of the staged class's working copy above
    b->fast = b->binom && elimination == BBX_GEBAUERMOELLER && sort_reducers && lg <= 256;
    b->fast_G = std::min(c.lds_max_basis ? lg : 256, c.max_basis);
    b->fast_P = std::min(2 * b->fast_G, c.max_pairs);
  }
  // non-binomial random ideals in <= 7 variables: wave-per-environment kernel, long-polynomial environments continue one
  // workgroup each (bbx_wide.h) behind it
  b->gen_to_wide = !b->binom && !b->wide && !b->staged && !b->fixed && !list && c.wide_waves >= 0;
  // (the counter of a two-kernel wide launch, BbxParams::wide_tail: allocated here, never inside a launch — a launch may be
  // recorded into a HIP graph)
  if ((b->wide || b->gen_to_wide) && b->ncu > 0 && batch > b->ncu) HIPCHK(hipMalloc((void**)&b->d_wide_done, 256));
  if (c.max_basis > 65535 || c.max_poly_terms > (1 << 22) || c.max_basis < 2 || c.max_pairs < 2 || c.max_poly_terms < 4 || c.queue_slots < 1)
    return fail(BBX_E_ARG, "capacities out of range");
  b->L = b->binom ? make_layout_binom(b->W, c.max_basis, c.max_pairs)
                  : make_layout(b->W, c.max_basis, c.max_pairs, c.arena_terms, c.max_poly_terms);
  // random distributions: the ideals are drawn on the device (same seeded streams; see gen_binomial / gen_polynomial in
  // bbx_device.h) and the ideal queue shrinks to one unused slot.  sort_input: the device sorts up to 16 generators
  // (gen_sorted_rank; std::sort is a stable insertion sort up to there, beyond it the host's std::sort decides ties).  Not
  // for ideal lists.
  std::vector<uint32_t> gen_table;
  if (!b->fixed && !list && !(sort_input && proto->npolys() > 16) && !getenv("BBX_HOST_GEN")) proto->device_table(b->W, &gen_table);
  b->nslots = (b->fixed || !gen_table.empty()) ? 1 : (uint32_t)c.queue_slots;
  b->slot_words = 1 + (uint32_t)proto->npolys() * (2 + (uint32_t)std::min(proto->max_terms_hint(), c.max_poly_terms) * (1 + b->W));
  b->slot_words = (b->slot_words + 3u) & ~3u;

  // An unseeded generator of the reference seeds itself from std::random_device (ideals.cpp:163-164, 209-210), so two
  // environments built without seed() see different ideals; same here: environment e starts from base + e with a
  // random base per handle (BBX_DEFAULT_SEED pins it for debugging).  bbx_seed makes a run reproducible.
  long long seed_base;
  if (const char* sb = getenv("BBX_DEFAULT_SEED")) seed_base = atoll(sb);
  else { std::random_device rd; seed_base = (long long)(rd() & 0x3fffffffu); }
  b->value_rng.seed((uint64_t)seed_base * 0x9E3779B97F4A7C15ull + 0x5851F42D4C957F2Dull);
  if (b->fixed) b->gens.push_back(std::move(proto));
  else {
    for (int e = 0; e < batch; e++) {
      if (list) b->gens.push_back(bbx::make_list(list, e, batch, proto->nvars()));   // environment e: ideals e, e+B, ...
      else { b->gens.push_back(proto->clone()); b->gens.back()->seed(seed_base + e); }
    }
  }
  const size_t qwords = b->fixed ? b->slot_words : (size_t)batch * b->nslots * b->slot_words;
  b->h_q.assign(qwords, 0u);
  b->h_tail.assign(batch, 0); b->h_head.assign(batch, 0);
  if (b->fixed) {
    bbx::HIdeal F; std::string err;
    b->gens[0]->next(F, &err);
    int rc = pack_ideal(b.get(), F, b->h_q.data());
    if (rc) return rc;
  }

  HIPCHK(hipMalloc((void**)&b->d_recs, (size_t)batch * b->L.rec_bytes));
  HIPCHK(hipMalloc((void**)&b->d_q, qwords * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void**)&b->d_tail, (size_t)batch * sizeof(int32_t)));
  { int rc_ = alloc_io(b.get(), batch); if (rc_) return rc_; }
  HIPCHK(hipMalloc((void**)&b->d_actions, (size_t)batch * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&b->d_mask, (size_t)batch));
  HIPCHK(hipMalloc((void**)&b->d_seeds, (size_t)batch * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void**)&b->d_hdr, (size_t)batch * sizeof(BbxHdr)));
  b->d_inv = inv_table(device);
  if (!b->d_inv) return fail(BBX_E_DEVICE, "no room for the inverse table");
  int lrc = bbx_launch_init(b->d_recs, b->L.rec_bytes, batch, nullptr, 0);
  if (lrc) return fail(BBX_E_DEVICE, "init launch failed: %s", hipGetErrorString((hipError_t)lrc));
  if (!gen_table.empty()) {
    HIPCHK(hipMalloc((void**)&b->d_gen, gen_table.size() * sizeof(uint32_t)));
    b->gen_owner = std::shared_ptr<uint32_t>(b->d_gen, [](uint32_t* q) { (void)hipFree(q); });
    HIPCHK(hipMemcpy(b->d_gen, gen_table.data(), gen_table.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    b->device_gen = true; b->gen_words = gen_table.size();
    std::vector<long long> seeds(batch);
    for (int e = 0; e < batch; e++) seeds[e] = seed_base + e;       // the default seeding of the host generators above
    HIPCHK(hipDeviceSynchronize());
    int rc = write_gen_states(b.get(), seeds);
    if (rc) return rc;
  }
  lrc = bbx_launch_mark_reset(b->d_recs, b->L.rec_bytes, batch, nullptr, 0);
  if (lrc) return fail(BBX_E_DEVICE, "init launch failed: %s", hipGetErrorString((hipError_t)lrc));
  HIPCHK(hipDeviceSynchronize());
  *out = b.release();
  return BBX_OK;
}

}  // namespace bbx_host
using namespace bbx_host;

extern "C" {

const char* bbx_last_error(void) { return g_err.c_str(); }
const char* bbx_version(void) { return "bbx 0.1 (gfx950)"; }
uint32_t bbx_agent_hash(uint32_t seed, uint32_t t) { return bbx_agent_hash32(seed, t); }
uint32_t bbx_agent_action(uint32_t seed, uint32_t t, uint32_t rows) { return bbx_agent_action32(seed, t, rows); }

int bbx_create(const char* ideal_dist, int elimination, int rewards, int sort_input, int sort_reducers,
               int k, int batch, int device, const bbx_caps* caps, bbx_batch** out) {
  if (!ideal_dist) return fail(BBX_E_ARG, "ideal_dist is null");
  std::string err;
  auto g = bbx::parse_ideal_dist(ideal_dist, &err);
  if (!g) return fail(BBX_E_ARG, "%s", err.c_str());
  return create_common(std::move(g), 0, elimination, rewards, sort_input, sort_reducers, k, batch, device, caps, out);
}

int bbx_create_fixed(int npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps, int nvars_obs,
                     int elimination, int rewards, int sort_input, int sort_reducers,
                     int k, int batch, int device, const bbx_caps* caps, bbx_batch** out) {
  if (npolys < 1 || !nterms || !coefs || !exps) return fail(BBX_E_ARG, "bad fixed ideal");
  bbx::HIdeal F;
  size_t at = 0;
  for (int p = 0; p < npolys; p++) {
    std::vector<bbx::HTerm> ts;
    for (int t = 0; t < nterms[p]; t++, at++) {
      bbx::HTerm h; h.c = bbx::coef_norm(coefs[at]); h.deg = 0;
      for (int v = 0; v < bbx::kN; v++) { h.e[v] = exps[at * bbx::kN + v]; h.deg += h.e[v]; }
      ts.push_back(h);
    }
    if (ts.empty()) return fail(BBX_E_ARG, "zero polynomial among the generators");
    F.push_back(bbx::poly_from_terms(ts));
  }
  return create_common(bbx::make_fixed(F), nvars_obs, elimination, rewards, sort_input, sort_reducers, k, batch, device, caps, out);
}

int bbx_create_ideals(int nideals, const int32_t* npolys, const int32_t* nterms, const int32_t* coefs, const int32_t* exps,
                      int nvars_obs, int elimination, int rewards, int sort_input, int sort_reducers,
                      int k, int batch, int device, const bbx_caps* caps, bbx_batch** out) {
  if (nideals < 1 || !npolys || !nterms || !coefs || !exps || nvars_obs < 1) return fail(BBX_E_ARG, "bad ideal list");
  auto list = std::make_shared<std::vector<bbx::HIdeal>>();
  size_t pi = 0, at = 0;
  for (int i = 0; i < nideals; i++) {
    bbx::HIdeal F;
    for (int p = 0; p < npolys[i]; p++, pi++) {
      std::vector<bbx::HTerm> ts;
      for (int t = 0; t < nterms[pi]; t++, at++) {
        bbx::HTerm h; h.c = bbx::coef_norm(coefs[at]); h.deg = 0;
        for (int v = 0; v < bbx::kN; v++) { h.e[v] = exps[at * bbx::kN + v]; h.deg += h.e[v]; }
        ts.push_back(h);
      }
      if (ts.empty()) return fail(BBX_E_ARG, "zero polynomial among the generators");
      F.push_back(bbx::poly_from_terms(ts));
    }
    if (F.empty()) return fail(BBX_E_ARG, "empty ideal in the list");
    list->push_back(F);
  }
  std::shared_ptr<const std::vector<bbx::HIdeal>> clist = list;
  return create_common(bbx::make_list(clist, 0, 1, nvars_obs), nvars_obs, elimination, rewards, sort_input, sort_reducers,
                       k, batch, device, caps, out, clist);
}

void bbx_destroy(bbx_batch* b) { delete b; }

int bbx_copy(const bbx_batch* s, bbx_batch** out) {
  if (!s || !out) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(s->device)); const_cast<bbx_batch*>(s)->api_epoch++;
  if (s->ps_active) { int rc_ = session_close(const_cast<bbx_batch*>(s), false, nullptr, false); if (rc_) return rc_; }
  HIPCHK(hipDeviceSynchronize());
  auto b = std::make_unique<bbx_batch>();
  b->B = s->B; b->device = s->device; b->k = s->k; b->nvars = s->nvars; b->W = s->W;
  b->ncu = s->ncu;
  b->elim = s->elim; b->rewards = s->rewards; b->sort_input = s->sort_input; b->sort_reducers = s->sort_reducers;
  b->fixed = s->fixed; b->listed = s->listed; b->binom = s->binom; b->L = s->L; b->LL = s->LL; b->slot_words = s->slot_words; b->nslots = s->nslots;
  b->h_q = s->h_q; b->h_tail = s->h_tail; b->h_head = s->h_head; b->q_dirty = true;
  b->no_growth = s->no_growth; b->value_rng = s->value_rng; b->gen_to_wide = s->gen_to_wide;
  if (s->d_wide_done) HIPCHK(hipMalloc((void**)&b->d_wide_done, 256));
  b->wide = s->wide; b->wide_terms = s->wide_terms; b->accounting = s->accounting; b->staged = s->staged; b->fast = s->fast; b->envs_per_block = s->envs_per_block;
  b->fast_G = s->fast_G; b->fast_P = s->fast_P;
  if (s->device_gen) {
    b->gen_owner = s->gen_owner; b->d_gen = s->d_gen;      // (immutable: shared)
    b->device_gen = true; b->gen_words = s->gen_words;
  }
  for (auto& g : s->gens) b->gens.push_back(g->clone());
  b->gen_error = s->gen_error;
  const int batch = s->B;
  HIPCHK(hipMalloc((void**)&b->d_recs, (size_t)batch * b->L.rec_bytes));
  HIPCHK(hipMemcpy(b->d_recs, s->d_recs, (size_t)batch * b->L.rec_bytes, hipMemcpyDeviceToDevice));
  HIPCHK(hipMalloc((void**)&b->d_q, b->h_q.size() * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void**)&b->d_tail, (size_t)batch * sizeof(int32_t)));
  { int rc_ = alloc_io(b.get(), batch); if (rc_) return rc_; }
  HIPCHK(hipMalloc((void**)&b->d_actions, (size_t)batch * sizeof(int32_t)));
  HIPCHK(hipMalloc((void**)&b->d_mask, (size_t)batch));
  HIPCHK(hipMalloc((void**)&b->d_seeds, (size_t)batch * sizeof(uint32_t)));
  HIPCHK(hipMalloc((void**)&b->d_hdr, (size_t)batch * sizeof(BbxHdr)));
  b->d_inv = s->d_inv;
  // (ideals drawn on the device: the host-side queue is never read by a kernel, so there is nothing to upload — two
  // pageable host-to-device copies that were a third of a one-environment copy)
  if (b->device_gen) { b->q_dirty = false; }
  int rc = upload_queue(b.get());
  if (rc) return rc;
  *out = b.release();
  return BBX_OK;
}

// In-batch clones for tree search (mcts.py:89,96,147 call env.copy() per expanded node): environment src[i] is copied
// over environment dst[i] — device record, pending queued ideals and the host generator's RNG state — without any
// allocation.  src and dst must not overlap.
int bbx_clone_envs(bbx_batch* b, int n, const int32_t* src, const int32_t* dst) {
  if (!b || n < 0 || (n && (!src || !dst))) return fail(BBX_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
  {
    std::vector<uint8_t> role((size_t)b->B, 0);          // 1: read, 2: written (a destination may be named once)
    for (int i = 0; i < n; i++) {
      if (src[i] < 0 || src[i] >= b->B || dst[i] < 0 || dst[i] >= b->B) return fail(BBX_E_ARG, "environment index out of range");
      role[src[i]] |= 1;
    }
    for (int i = 0; i < n; i++) {
      if (role[dst[i]] & 1) return fail(BBX_E_ARG, "source and destination sets overlap");
      if (role[dst[i]] & 2) return fail(BBX_E_ARG, "environment %d is the destination of two clones", dst[i]);
      role[dst[i]] |= 2;
    }
  }
  if (n == 0) return BBX_OK;
  int rc = read_lite(b, 0);                       // current queue heads
  if (rc) return rc;
  if (b->clone_cap < n) {                         // index arrays of the clone kernel: kept with the handle
    if (b->d_clone_idx) (void)hipFree(b->d_clone_idx);
    b->d_clone_idx = nullptr; b->clone_cap = 0;
    HIPCHK(hipMalloc((void**)&b->d_clone_idx, (size_t)2 * n * sizeof(int32_t)));
    b->clone_cap = n;
  }
  int32_t* d_s = b->d_clone_idx; int32_t* d_d = b->d_clone_idx + n;
  HIPCHK(hipMemcpyAsync(d_s, src, (size_t)n * 4, hipMemcpyHostToDevice, nullptr));
  HIPCHK(hipMemcpyAsync(d_d, dst, (size_t)n * 4, hipMemcpyHostToDevice, nullptr));
  int lrc = bbx_launch_clone(b->d_recs, b->d_recs, &b->L, d_s, d_d, n, nullptr, 1, 0, 0, nullptr, 0);
  HIPCHK(hipStreamSynchronize(nullptr));
  if (lrc) return fail(BBX_E_DEVICE, "clone launch failed: %s", hipGetErrorString((hipError_t)lrc));
  if (b->device_gen) return BBX_OK;                // (the generator's state is a header field: it travelled with the record)
  if (!b->fixed) {
    const size_t stride = (size_t)b->nslots * b->slot_words;
    if (b->q_dirty_env.size() != (size_t)b->B) b->q_dirty_env.assign(b->B, b->q_dirty ? 1 : 0);
    for (int i = 0; i < n; i++) {
      const int s = src[i], d = dst[i];
      b->gens[d] = b->gens[s]->clone();
      if (!b->gen_error.empty()) b->gen_error[d] = b->gen_error[s];
      b->h_tail[d] = b->h_tail[s]; b->h_head[d] = b->h_head[s];
      memcpy(b->h_q.data() + (size_t)d * stride, b->h_q.data() + (size_t)s * stride, stride * sizeof(uint32_t));
      b->q_dirty_env[d] = 1;
    }
    b->q_dirty = true;
    return upload_queue(b);
  }
  return BBX_OK;
}

int bbx_seed(bbx_batch* b, const int64_t* seeds) {
  if (!b || !seeds) return fail(BBX_E_ARG, "null argument");
  if (b->fixed) return BBX_OK;                 // FixedIdealGenerator::seed is a no-op (ideals.h:94)
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->device_gen) {
    if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
    std::vector<long long> s(seeds, seeds + b->B);
    for (int e = 0; e < b->B; e++) b->gens[e]->seed(seeds[e]);     // (kept in step for bbx_copy of a host-generating twin)
    return write_gen_states(b, s);
  }
  int rc = read_headers(b);                    // ideals generated ahead from the old stream are dropped
  if (rc) return rc;
  for (int e = 0; e < b->B; e++) { b->gens[e]->seed(seeds[e]); b->h_tail[e] = b->h_head[e]; }
  b->gen_error.clear();
  b->q_dirty = true; b->q_dirty_env.clear();
  return BBX_OK;
}

int bbx_seed_agent(bbx_batch* b, const uint32_t* seeds) {
  if (!b || !seeds) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->in_flight) { int rc_ = finish(b, b->last_stream); if (rc_) return rc_; }   // (the headers are the truth only when nothing is resident)
  // written straight into the headers (field agent_seed), strided copy
  HIPCHK(hipMemcpy2D(b->d_recs + offsetof(BbxHdr, agent_seed), b->L.rec_bytes, seeds, sizeof(uint32_t), sizeof(uint32_t), b->B, hipMemcpyHostToDevice));
  // the agent's step counter restarts with a new seed
  std::vector<int32_t> zero(b->B, 0);
  HIPCHK(hipMemcpy2D(b->d_recs + offsetof(BbxHdr, t), b->L.rec_bytes, zero.data(), sizeof(int32_t), sizeof(int32_t), b->B, hipMemcpyHostToDevice));
  return BBX_OK;
}

int bbx_seed_strategy(bbx_batch* b, const int64_t* seeds) {
  if (!b || !seeds) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
  // linear_congruential_engine<uint_fast32_t, 16807, 0, 2^31-1>::seed(s): the int seed converts to the 64-bit
  // unsigned result_type first; x = s mod m, and 0 becomes 1 (libstdc++-11 bits/random.tcc)
  std::vector<uint32_t> st(b->B);
  for (int e = 0; e < b->B; e++) {
    uint32_t x = (uint32_t)((uint64_t)seeds[e] % 2147483647ull);
    st[e] = x ? x : 1u;
  }
  HIPCHK(hipMemcpy2D(b->d_recs + offsetof(BbxHdr, std_rng), b->L.rec_bytes, st.data(), sizeof(uint32_t), sizeof(uint32_t), b->B, hipMemcpyHostToDevice));
  return BBX_OK;
}

int bbx_reset(bbx_batch* b, const uint8_t* mask, int32_t* rows) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  // (a session's waves hold their environments in registers: the mark below must meet the records they have stored)
  if (b->in_flight) { int rc_ = finish(b, b->last_stream); if (rc_) return rc_; }
  if (mask) HIPCHK(hipMemcpy(b->d_mask, mask, (size_t)b->B, hipMemcpyHostToDevice));
  int lrc = bbx_launch_mark_reset(b->d_recs, b->L.rec_bytes, b->B, mask ? b->d_mask : nullptr, 0);
  if (lrc) return fail(BBX_E_DEVICE, "launch failed: %s", hipGetErrorString((hipError_t)lrc));
  BbxParams p; fill_params(b, &p);
  p.nsteps = 0; p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = 0;
  p.rewards = b->d_rewards; p.dones = b->d_dones; p.rows = b->d_rows;
  int rc = launch(b, p, 0);
  if (rc) return rc;
  rc = finish(b, 0);
  if (rc) return rc;
  if (rows) for (int e = 0; e < b->B; e++) rows[e] = b->h_lite[(size_t)e * 4 + 3];
  return BBX_OK;
}

static int step_host(bbx_batch* b, const int32_t* actions, double* rewards, uint8_t* dones, int32_t* rows, int auto_reset) {
  if (!b || !actions) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  memcpy(b->h_act, actions, (size_t)b->B * sizeof(int32_t));
  if (mbox_eligible(b)) {
    BbxParams p; fill_params(b, &p);
    p.nsteps = 1; p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = auto_reset;
    p.actions = b->zc_act_dev; zc_outputs(b, &p);
    bool used = false;
    int rc = mbox_step(b, p, &used);
    if (rc) return rc;
    if (used) { b->mbox_epoch = b->api_epoch; return copy_out(b, rewards, dones, rows); }
  }
  BbxParams p; fill_params(b, &p);
  p.nsteps = 1; p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = auto_reset;
  if (b->zero_copy) { p.actions = b->zc_act_dev; zc_outputs(b, &p); b->zc_active = true; }
  else {
    HIPCHK(hipMemcpyAsync(b->d_actions, b->h_act, (size_t)b->B * sizeof(int32_t), hipMemcpyHostToDevice, 0));
    p.actions = b->d_actions; p.rewards = b->d_rewards; p.dones = b->d_dones; p.rows = b->d_rows;
  }
  int rc = launch(b, p, 0);
  if (!rc) rc = finish(b, 0);
  b->zc_active = false;
  if (rc) return rc;
  b->mbox_epoch = b->api_epoch;
  return copy_out(b, rewards, dones, rows);
}

int bbx_step(bbx_batch* b, const int32_t* actions, double* rewards, uint8_t* dones, int32_t* rows) {
  return step_host(b, actions, rewards, dones, rows, 0);
}
int bbx_step_autoreset(bbx_batch* b, const int32_t* actions, double* rewards, uint8_t* dones, int32_t* rows) {
  return step_host(b, actions, rewards, dones, rows, 1);
}

// (re)size the padded device observation block
static int ensure_obs_block(bbx_batch* b, int rows_cap) {
  if (b->obs_rows_cap >= (size_t)rows_cap) return BBX_OK;
  const size_t cols = (size_t)2 * b->nvars * b->k;
  if (b->d_obs) HIPCHK(hipFree(b->d_obs));
  b->d_obs = nullptr; b->obs_rows_cap = 0;
  HIPCHK(hipMalloc((void**)&b->d_obs, (size_t)b->B * rows_cap * cols * sizeof(int32_t)));
  b->obs_rows_cap = rows_cap;
  return BBX_OK;
}

int bbx_step_obs(bbx_batch* b, const int32_t* actions, int auto_reset, double* rewards, uint8_t* dones, int32_t* rows,
                 const int32_t** obs, const int32_t** offsets) {
  if (!b || !obs || !offsets) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  const int cols = 2 * b->nvars * b->k;
  const bool zc = b->zero_copy;
  int rc;
  if (zc) {
    if (!b->h_zobs) {
      b->zobs_rows_cap = 128;
      HIPCHK(hipHostMalloc((void**)&b->h_zobs, (size_t)b->B * b->zobs_rows_cap * cols * sizeof(int32_t), hipHostMallocCoherent | hipHostMallocMapped));
      HIPCHK(hipHostGetDevicePointer((void**)&b->zc_obs_dev, b->h_zobs, 0));
    }
  } else {
    rc = ensure_obs_block(b, b->obs_rows_cap ? (int)b->obs_rows_cap : 128);
    if (rc) return rc;
    if (!b->d_obs_off) HIPCHK(hipMalloc((void**)&b->d_obs_off, ((size_t)b->B + 1) * sizeof(int32_t)));
  }
  if (actions) {
    memcpy(b->h_act, actions, (size_t)b->B * sizeof(int32_t));
    if (!zc) HIPCHK(hipMemcpyAsync(b->d_actions, b->h_act, (size_t)b->B * sizeof(int32_t), hipMemcpyHostToDevice, 0));
  }
  for (int attempt = 0;; attempt++) {
    // (the parameters are formed per attempt: the first one may have enlarged the records — another array, another layout —
    // and a second attempt with the first one's parameters stepped the freed copy: found by scripts/fuzz_gym.py, round 4)
    BbxParams p; fill_params(b, &p);
    p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = auto_reset ? 1 : 0;
    if (zc) zc_outputs(b, &p); else { p.rewards = b->d_rewards; p.dones = b->d_dones; p.rows = b->d_rows; }
    if (actions && !attempt) { p.actions = zc ? b->zc_act_dev : b->d_actions; p.nsteps = 1; }
    else { p.nsteps = 0; p.actions = nullptr; }             // observation of the current state only / the step is done: only rewrite the observation
    const size_t cap = zc ? b->zobs_rows_cap : b->obs_rows_cap;
    p.obs = zc ? b->zc_obs_dev : b->d_obs; p.obs_rows = (int)cap; p.obs_fill = 0; p.obs_every_step = 0;
    bool used = false;
    if (!attempt && actions && zc && mbox_eligible(b)) {    // a step of a loop: through the resident kernel's mailbox
      BbxParams q = p;
      q.obs_every_step = 1;                                 // (the kernel stays: every step writes its observation)
      rc = mbox_step(b, q, &used);
      if (rc) return rc;
    }
    if (!used) {
      b->zc_active = zc;
      rc = launch(b, p, 0);
      if (!rc) rc = finish(b, 0);
      b->zc_active = false;
      if (rc) return rc;
    }
    if (!attempt) copy_out(b, rewards, dones, rows);
    int maxr = 0; size_t total = 0;
    for (int e = 0; e < b->B; e++) { const int r = b->h_lite[(size_t)e * 4 + 3]; maxr = r > maxr ? r : maxr; total += (size_t)r; }
    if ((size_t)maxr > cap) {                               // some pair set outgrew the block: enlarge it and write again
      size_t ncap = cap;
      while (ncap < (size_t)maxr) ncap *= 2;
      if (zc) {
        (void)hipHostFree(b->h_zobs); b->h_zobs = nullptr;
        HIPCHK(hipHostMalloc((void**)&b->h_zobs, (size_t)b->B * ncap * cols * sizeof(int32_t), hipHostMallocCoherent | hipHostMallocMapped));
        HIPCHK(hipHostGetDevicePointer((void**)&b->zc_obs_dev, b->h_zobs, 0));
        b->zobs_rows_cap = ncap;
      } else {
        rc = ensure_obs_block(b, (int)ncap);
        if (rc) return rc;
      }
      continue;
    }
    const size_t need = (total ? total : 1) * (size_t)cols + (size_t)b->B + 1;
    if (need > b->obs_packed_cap) {
      if (b->d_obs_packed) (void)hipFree(b->d_obs_packed);
      if (b->h_obs) (void)hipHostFree(b->h_obs);
      b->d_obs_packed = nullptr; b->h_obs = nullptr; b->obs_packed_cap = 0;
      if (!zc) HIPCHK(hipMalloc((void**)&b->d_obs_packed, need * 2 * sizeof(int32_t)));
      HIPCHK(hipHostMalloc((void**)&b->h_obs, need * 2 * sizeof(int32_t), hipHostMallocDefault));
      b->obs_packed_cap = need * 2;
    }
    // pinned layout: [B + 1 offsets (rows)] [total * cols values]; the offsets are a host-side prefix sum of the rows
    // just fetched (the device computes the same ones for its pack kernel)
    b->h_obs[0] = 0;
    for (int e = 0; e < b->B; e++) b->h_obs[e + 1] = b->h_obs[e] + b->h_lite[(size_t)e * 4 + 3];
    if (total && zc) {                                      // the padded block is already in host memory: squeeze it here
      for (int e = 0; e < b->B; e++)
        memcpy(b->h_obs + b->B + 1 + (size_t)b->h_obs[e] * cols, b->h_zobs + (size_t)e * cap * cols,
               (size_t)(b->h_obs[e + 1] - b->h_obs[e]) * cols * sizeof(int32_t));
    } else if (total) {
      const int32_t* src = b->d_obs;                         // one environment: its padded block IS the ragged one
      if (b->B > 1) {
        int lrc = bbx_launch_obs_pack(b->d_obs, (int)b->obs_rows_cap, cols, b->d_rows, b->B, b->d_obs_off, b->d_obs_packed, 0);
        if (lrc) return fail(BBX_E_DEVICE, "observation pack launch failed: %s", hipGetErrorString((hipError_t)lrc));
        src = b->d_obs_packed;
      }
      HIPCHK(hipMemcpyAsync(b->h_obs + b->B + 1, src, total * cols * sizeof(int32_t), hipMemcpyDeviceToHost, 0));
      HIPCHK(hipStreamSynchronize(0));
    }
    *offsets = b->h_obs; *obs = b->h_obs + b->B + 1;
    if (actions) b->mbox_epoch = b->api_epoch;              // (a step: the next one in a row counts towards a mailbox session)
    return BBX_OK;
  }
}

int bbx_rollout(bbx_batch* b, int agent, int nsteps, int auto_reset, double* rewards, uint8_t* dones, int32_t* rows) {
  if (!b || nsteps < 0 || agent < BBX_RANDOM_HASH || agent > BBX_RANDOM_STD) return fail(BBX_E_ARG, "bad rollout arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->d_trace && nsteps > b->trace_cap) return fail(BBX_E_ARG, "rollout of %d steps exceeds the trace capacity %d", nsteps, b->trace_cap);
  BbxParams p; fill_params(b, &p);
  p.nsteps = nsteps; p.set_budget = 1; p.agent = agent; p.auto_reset = auto_reset ? 1 : 0;
  p.rewards = b->d_rewards; p.dones = b->d_dones; p.rows = b->d_rows;
  int rc = launch(b, p, 0);
  if (rc) return rc;
  rc = finish(b, 0);
  if (rc) return rc;
  return copy_out(b, rewards, dones, rows);
}

static int step_device(bbx_batch* b, const int32_t* d_actions, double* d_rewards, uint8_t* d_dones, int32_t* d_rows,
                       int32_t* d_obs, int obs_rows, int obs_fill, void* stream, int auto_reset) {
  if (!b || !d_actions) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (d_obs && obs_rows < 1) return fail(BBX_E_ARG, "obs_rows must be positive");
  BbxParams p; fill_params(b, &p);
  p.nsteps = 1; p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = auto_reset; p.actions = d_actions;
  p.rewards = d_rewards; p.dones = d_dones; p.rows = d_rows; p.obs = d_obs; p.obs_rows = obs_rows; p.obs_fill = obs_fill;
  if (b->d_trace && b->trace_cap < 1) p.trace = nullptr;
  return launch(b, p, (hipStream_t)stream, d_obs != nullptr, true);
}

int bbx_step_device(bbx_batch* b, const int32_t* d_actions, double* d_rewards, uint8_t* d_dones, int32_t* d_rows,
                    int32_t* d_obs, int obs_rows, int obs_fill, void* stream) {
  return step_device(b, d_actions, d_rewards, d_dones, d_rows, d_obs, obs_rows, obs_fill, stream, 0);
}
int bbx_step_device_autoreset(bbx_batch* b, const int32_t* d_actions, double* d_rewards, uint8_t* d_dones, int32_t* d_rows,
                              int32_t* d_obs, int obs_rows, int obs_fill, void* stream) {
  return step_device(b, d_actions, d_rewards, d_dones, d_rows, d_obs, obs_rows, obs_fill, stream, 1);
}

// prepared-weights geometry: the same constexpr rules as bbx_pmlp.h (k-steps 3 / 6 / 10 / 16 / 32, unit blocks 1 / 2 / 4 / 8)
static int pmlp_ks(int cols) { const int ks = (cols + 1) / 2; return ks <= 3 ? 3 : ks <= 6 ? 6 : ks <= 10 ? 10 : ks <= 16 ? 16 : 32; }
static int pmlp_nb(int hidden) { const int nb = (hidden + 31) / 32; return nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : 8; }

int bbx_pmlp_prepared_floats(int cols, int hidden) {
  if (cols < 1 || cols > 64 || hidden < 1 || hidden > 256) return fail(BBX_E_UNSUPPORTED, "policy shape %d x %d is not built into the policy kernel", cols, hidden);
  return (2 * pmlp_ks(cols) + 2) * 32 * pmlp_nb(hidden) + 4;
}

int bbx_pmlp_prepare(const float* d_w1, const float* d_b1, const float* d_w2, float b2, int cols, int hidden, float* d_prepared, void* stream) {
  if (!d_w1 || !d_b1 || !d_w2 || !d_prepared) return fail(BBX_E_ARG, "null argument");
  if (bbx_pmlp_prepared_floats(cols, hidden) < 0) return BBX_E_UNSUPPORTED;
  int lrc = bbx_launch_pmlp_prepare(d_w1, d_b1, d_w2, b2, cols, hidden, d_prepared, (hipStream_t)stream);
  if (lrc) return fail(BBX_E_DEVICE, "policy launch failed: %s", hipGetErrorString((hipError_t)lrc));
  return BBX_OK;
}

int bbx_pmlp_act(const int32_t* d_obs, const int32_t* d_rows, int batch, int obs_rows, int cols, const float* d_prepared, int hidden,
                 const float* d_u, int32_t* d_actions, float* d_logprobs, void* stream) {
  if (!d_obs || !d_rows || !d_prepared || !d_u || !d_actions || !d_logprobs) return fail(BBX_E_ARG, "null argument");
  if (batch < 1 || obs_rows < 1) return fail(BBX_E_ARG, "bad policy shape");
  if (obs_rows > BBX_POLICY_MAX_ROWS) return fail(BBX_E_UNSUPPORTED, "the policy kernels score at most %d rows per environment (obs_rows = %d)", BBX_POLICY_MAX_ROWS, obs_rows);
  if (bbx_pmlp_prepared_floats(cols, hidden) < 0) return BBX_E_UNSUPPORTED;
  int lrc = bbx_launch_pmlp_act(d_obs, d_rows, batch, obs_rows, cols, d_prepared, hidden, d_u, d_actions, d_logprobs, (hipStream_t)stream);
  if (lrc) return fail(BBX_E_DEVICE, "policy launch failed: %s", hipGetErrorString((hipError_t)lrc));
  return BBX_OK;
}

// ---- two and three hidden layers (bbx_pmlp2.hip; hm = 0: no middle layer)
extern "C" int bbx_pmlp2_floats(int cols, int h1, int hm, int h2);
extern "C" int bbx_launch_pmlp2_prepare(const float* w1, const float* b1, const float* wm, const float* bm, const float* w2, const float* b2,
                                        const float* wd, const float* bd, int cols, int h1, int hm, int h2, float* out, hipStream_t stream);
extern "C" int bbx_launch_pmlp2_act(const int32_t* obs, const int32_t* rows, int B, int obs_rows, int cols, const float* wp, int h1, int hm, int h2,
                                    const float* u, int32_t* actions, float* logprobs, int cus, int max_lds, hipStream_t stream);

static int pmlp_deep_floats(int cols, int h1, int hm, int h2, bool three) {
  if (cols < 1 || cols > 64 || h1 < 1 || h1 > 128 || h2 < 1 || h2 > 128 || (three && (hm < 1 || hm > 128)))
    return three ? fail(BBX_E_UNSUPPORTED, "policy shape %d x %d x %d x %d is not built into the three-layer policy kernel", cols, h1, hm, h2)
                 : fail(BBX_E_UNSUPPORTED, "policy shape %d x %d x %d is not built into the two-layer policy kernel", cols, h1, h2);
  return bbx_pmlp2_floats(cols, h1, three ? hm : 0, h2);
}
static int pmlp_deep_act(const int32_t* d_obs, const int32_t* d_rows, int batch, int obs_rows, int cols, const float* d_prepared, int h1, int hm, int h2,
                         bool three, const float* d_u, int32_t* d_actions, float* d_logprobs, void* stream) {
  if (!d_obs || !d_rows || !d_prepared || !d_u || !d_actions || !d_logprobs) return fail(BBX_E_ARG, "null argument");
  if (batch < 1 || obs_rows < 1) return fail(BBX_E_ARG, "bad policy shape");
  if (obs_rows > BBX_POLICY_MAX_ROWS) return fail(BBX_E_UNSUPPORTED, "the policy kernels score at most %d rows per environment (obs_rows = %d)", BBX_POLICY_MAX_ROWS, obs_rows);
  if (pmlp_deep_floats(cols, h1, hm, h2, three) < 0) return BBX_E_UNSUPPORTED;
  int dev = 0, cus = 0, max_lds = 0;
  HIPCHK(hipGetDevice(&dev));
  {
    // per device, asked once (this sits on the per-step path of a policy rollout).  gfx950 has 160 KB of LDS per workgroup, whatever
    // hipDeviceAttributeMaxSharedMemoryPerBlock says (64 KB: the limit without the per-function attribute); the library is built
    // for that part only, so that figure is the floor for it and for nothing else
    static std::mutex mu; static int c_cus[64], c_lds[64]; static bool c_have[64];
    std::lock_guard<std::mutex> g(mu);
    const int d = dev & 63;
    if (!c_have[d]) {
      HIPCHK(hipDeviceGetAttribute(&c_cus[d], hipDeviceAttributeMultiprocessorCount, dev));
      HIPCHK(hipDeviceGetAttribute(&c_lds[d], hipDeviceAttributeMaxSharedMemoryPerBlock, dev));
      hipDeviceProp_t prop;
      HIPCHK(hipGetDeviceProperties(&prop, dev));
      if (strncmp(prop.gcnArchName, "gfx950", 6) == 0 && c_lds[d] < 163840) c_lds[d] = 163840;
      c_have[d] = true;
    }
    cus = c_cus[d]; max_lds = c_lds[d];
  }
  int lrc = bbx_launch_pmlp2_act(d_obs, d_rows, batch, obs_rows, cols, d_prepared, h1, three ? hm : 0, h2, d_u, d_actions, d_logprobs, cus, max_lds,
                                 (hipStream_t)stream);
  if (lrc == (int)hipErrorInvalidValue)
    return fail(BBX_E_UNSUPPORTED, "the policy kernel needs more LDS than device %d has (%d bytes per workgroup)", dev, max_lds);
  if (lrc) return fail(BBX_E_DEVICE, "policy launch failed: %s", hipGetErrorString((hipError_t)lrc));
  return BBX_OK;
}

int bbx_pmlp2_prepared_floats(int cols, int hidden1, int hidden2) { return pmlp_deep_floats(cols, hidden1, 0, hidden2, false); }

int bbx_pmlp2_prepare(const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2, const float* d_w3, const float* d_b3,
                      int cols, int hidden1, int hidden2, float* d_prepared, void* stream) {
  if (!d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_w3 || !d_b3 || !d_prepared) return fail(BBX_E_ARG, "null argument");
  if (bbx_pmlp2_prepared_floats(cols, hidden1, hidden2) < 0) return BBX_E_UNSUPPORTED;
  int lrc = bbx_launch_pmlp2_prepare(d_w1, d_b1, nullptr, nullptr, d_w2, d_b2, d_w3, d_b3, cols, hidden1, 0, hidden2, d_prepared, (hipStream_t)stream);
  if (lrc) return fail(BBX_E_DEVICE, "policy launch failed: %s", hipGetErrorString((hipError_t)lrc));
  return BBX_OK;
}

int bbx_pmlp2_act(const int32_t* d_obs, const int32_t* d_rows, int batch, int obs_rows, int cols, const float* d_prepared, int hidden1, int hidden2,
                  const float* d_u, int32_t* d_actions, float* d_logprobs, void* stream) {
  return pmlp_deep_act(d_obs, d_rows, batch, obs_rows, cols, d_prepared, hidden1, 0, hidden2, false, d_u, d_actions, d_logprobs, stream);
}

int bbx_pmlp3_prepared_floats(int cols, int hidden1, int hidden2, int hidden3) { return pmlp_deep_floats(cols, hidden1, hidden2, hidden3, true); }

int bbx_pmlp3_prepare(const float* d_w1, const float* d_b1, const float* d_w2, const float* d_b2, const float* d_w3, const float* d_b3,
                      const float* d_w4, const float* d_b4, int cols, int hidden1, int hidden2, int hidden3, float* d_prepared, void* stream) {
  if (!d_w1 || !d_b1 || !d_w2 || !d_b2 || !d_w3 || !d_b3 || !d_w4 || !d_b4 || !d_prepared) return fail(BBX_E_ARG, "null argument");
  if (bbx_pmlp3_prepared_floats(cols, hidden1, hidden2, hidden3) < 0) return BBX_E_UNSUPPORTED;
  int lrc = bbx_launch_pmlp2_prepare(d_w1, d_b1, d_w2, d_b2, d_w3, d_b3, d_w4, d_b4, cols, hidden1, hidden2, hidden3, d_prepared, (hipStream_t)stream);
  if (lrc) return fail(BBX_E_DEVICE, "policy launch failed: %s", hipGetErrorString((hipError_t)lrc));
  return BBX_OK;
}

int bbx_pmlp3_act(const int32_t* d_obs, const int32_t* d_rows, int batch, int obs_rows, int cols, const float* d_prepared, int hidden1, int hidden2,
                  int hidden3, const float* d_u, int32_t* d_actions, float* d_logprobs, void* stream) {
  return pmlp_deep_act(d_obs, d_rows, batch, obs_rows, cols, d_prepared, hidden1, hidden2, hidden3, true, d_u, d_actions, d_logprobs, stream);
}

int bbx_policy_step_device(bbx_batch* b, const float* d_prepared, int hidden, const float* d_u, int32_t* d_actions, float* d_logprobs,
                           double* d_rewards, uint8_t* d_dones, int32_t* d_rows, int32_t* d_obs, int obs_rows, int obs_fill, void* stream) {
  if (!b || !d_prepared || !d_u || !d_actions || !d_logprobs || !d_rows || !d_obs) return fail(BBX_E_ARG, "null argument");
  if (obs_rows < 1) return fail(BBX_E_ARG, "obs_rows must be positive");
  if (obs_rows > BBX_POLICY_MAX_ROWS) return fail(BBX_E_UNSUPPORTED, "the policy kernels score at most %d rows per environment (obs_rows = %d)", BBX_POLICY_MAX_ROWS, obs_rows);
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  const int cols = 2 * b->nvars * b->k;
  if (bbx_pmlp_prepared_floats(cols, hidden) < 0) return BBX_E_UNSUPPORTED;
  // one launch for policy + step where the step kernel has the policy built in (the register/LDS-resident class, lean
  // variant, 33..128 hidden units, at most 12 columns); everywhere else the two launches it replaces
  const bool fused = b->fast && b->staged && !b->accounting && !(b->d_trace && b->trace_cap >= 1) && (pmlp_nb(hidden) == 2 || pmlp_nb(hidden) == 4) &&
                     cols <= 12 && !getenv("BBX_NO_FUSED_POLICY");
  if (!fused) {
    int rc = bbx_pmlp_act(d_obs, d_rows, b->B, obs_rows, cols, d_prepared, hidden, d_u, d_actions, d_logprobs, stream);
    if (rc) return rc;
    return step_device(b, d_actions, d_rewards, d_dones, d_rows, d_obs, obs_rows, obs_fill, stream, 1);
  }
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->ps_enabled && b->nvars == 3 && b->k == 2 && b->device_gen) {
    // persistent sessions: the call joins (or begins) a session whose kernel has the policy inside its step loop — the
    // uniforms of consecutive calls must then be consecutive [B] slices of one array (what a rollout loop that draws its
    // random numbers a chunk of steps at a time passes), every other argument the same from call to call
    BbxPolicy spol{d_prepared, hidden, d_u, d_actions, d_logprobs, 1, d_rewards, d_dones, d_rows, 0, 0, 1};
    BbxParams sp; fill_params(b, &sp);
    sp.nsteps = 1; sp.set_budget = 1; sp.agent = BBX_AGENT_EXTERNAL; sp.auto_reset = 1;
    sp.obs = d_obs; sp.obs_rows = obs_rows; sp.obs_fill = obs_fill; sp.trace = nullptr;
    sp.policy = &spol;
    return launch(b, sp, (hipStream_t)stream, true, true);
  }
  BbxPolicy pol{d_prepared, hidden, d_u, d_actions, d_logprobs, 0, nullptr, nullptr, nullptr, 0, 0, 0};
  BbxParams p; fill_params(b, &p);
  p.nsteps = 1; p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = 1; p.actions = d_actions;   // (the follow-up pass reads them)
  p.rewards = d_rewards; p.dones = d_dones; p.rows = d_rows; p.obs = d_obs; p.obs_rows = obs_rows; p.obs_fill = obs_fill;
  p.trace = nullptr;
  p.policy = &pol;
  return launch(b, p, (hipStream_t)stream, true, true);
}

int bbx_policy_rollout_device(bbx_batch* b, const float* d_prepared, int hidden, int nsteps, const float* d_u, int32_t* d_actions,
                              float* d_logprobs, double* d_rewards, uint8_t* d_dones, int32_t* d_rows, int32_t* d_obs, int obs_rows,
                              long long obs_step_stride, void* stream) {
  if (!b || !d_prepared || !d_u || !d_actions || !d_logprobs) return fail(BBX_E_ARG, "null argument");
  if (nsteps < 1 || (d_obs && obs_rows < 1) || obs_step_stride < 0) return fail(BBX_E_ARG, "bad rollout arguments");
  if (d_obs && obs_rows > BBX_POLICY_MAX_ROWS) return fail(BBX_E_UNSUPPORTED, "the policy kernels score at most %d rows per environment (obs_rows = %d)", BBX_POLICY_MAX_ROWS, obs_rows);
  const int cols = 2 * b->nvars * b->k;
  if (bbx_pmlp_prepared_floats(cols, hidden) < 0) return BBX_E_UNSUPPORTED;
  // where the policy is built into the step kernels: binomial classes with 8- or 16-byte monomials, 33..128 hidden units,
  // observation widths whose prepared weights have 6 k-steps (or 10 with 16-byte monomials)
  const int ks = pmlp_ks(cols);
  if (!b->binom || b->wide || (b->W != 2 && b->W != 4) || (pmlp_nb(hidden) != 2 && pmlp_nb(hidden) != 4) || !(ks == 6 || (b->W == 4 && ks == 10)))
    return fail(BBX_E_UNSUPPORTED, "policy rollouts are built into the binomial kernel classes only (<= 7 variables, 2nk <= 12 columns, or <= 20 with "
                                   "more than 3 variables; 33..128 hidden units); drive this batch with bbx_policy_step_device");
  if (b->accounting) return fail(BBX_E_UNSUPPORTED, "policy rollouts run the lean kernel: call bbx_accounting(b, 0) first");
  if (b->d_trace && b->trace_cap >= 1) return fail(BBX_E_UNSUPPORTED, "policy rollouts are not traced");
  if (d_obs && obs_step_stride != 0 && obs_step_stride < (long long)b->B * obs_rows * cols) return fail(BBX_E_ARG, "obs_step_stride smaller than one block");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  BbxPolicy pol{d_prepared, hidden, d_u, d_actions, d_logprobs, 1, d_rewards, d_dones, d_rows, obs_step_stride, b->B, 0};
  BbxParams p; fill_params(b, &p);
  p.nsteps = nsteps; p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = 1;
  p.obs = d_obs; p.obs_rows = d_obs ? obs_rows : 0; p.obs_fill = 0;   // (no block: the kernels size their logits for every row they score)
  p.trace = nullptr;
  p.policy = &pol;
  // the register/LDS-resident kernel has the policy for 3 variables and k = 2; every other admitted shape runs in the
  // HBM-resident binomial kernel from the start
  pol.rollout = (b->fast && b->staged && b->nvars == 3 && b->k == 2) ? 1 : 2;
  return launch(b, p, (hipStream_t)stream, true, true);   // (rows the policy could not score — more than the block or the kernel holds — are an error)
}

int bbx_rollout_device(bbx_batch* b, int agent, int nsteps, int auto_reset, double* d_rewards, uint8_t* d_dones,
                       int32_t* d_rows, int32_t* d_obs, int obs_rows, int obs_fill, int obs_every_step, void* stream) {
  if (!b || nsteps < 0 || agent < BBX_RANDOM_HASH || agent > BBX_RANDOM_STD) return fail(BBX_E_ARG, "bad rollout arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->d_trace && nsteps > b->trace_cap) return fail(BBX_E_ARG, "rollout of %d steps exceeds the trace capacity %d", nsteps, b->trace_cap);
  if (d_obs && obs_rows < 1) return fail(BBX_E_ARG, "obs_rows must be positive");
  BbxParams p; fill_params(b, &p);
  p.obs_every_step = obs_every_step ? 1 : 0;
  p.nsteps = nsteps; p.set_budget = 1; p.agent = agent; p.auto_reset = auto_reset ? 1 : 0;
  p.rewards = d_rewards; p.dones = d_dones; p.rows = d_rows; p.obs = d_obs; p.obs_rows = obs_rows; p.obs_fill = obs_fill;
  return launch(b, p, (hipStream_t)stream, d_obs != nullptr, true);
}

int bbx_sync(bbx_batch* b) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (!b->in_flight) { HIPCHK(hipDeviceSynchronize()); return BBX_OK; }
  return finish(b, b->last_stream);
}

int bbx_accounting(bbx_batch* b, int enable) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  b->accounting = enable != 0;
  return BBX_OK;
}

int bbx_prefetch(bbx_batch* b) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
  else { int rc = read_headers(b); if (rc) return rc; }
  return fill_queues(b, (int)b->nslots);
}

int bbx_timing(bbx_batch* b, int enable, double* kernel_ms, int32_t* launches) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  if (kernel_ms) *kernel_ms = b->kernel_ms;
  if (launches) *launches = b->kernel_launches;
  b->kernel_ms = 0.0; b->kernel_launches = 0;
  b->timing = enable != 0;
  return BBX_OK;
}

int bbx_obs(bbx_batch* b, int32_t* out, int max_rows, int fill) {
  if (!b || !out || max_rows < 1) return fail(BBX_E_ARG, "bad arguments");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  const size_t cols = (size_t)2 * b->nvars * b->k;
  const size_t need = (size_t)b->B * max_rows * cols;
  if (b->obs_rows_cap < (size_t)max_rows) {
    if (b->d_obs) HIPCHK(hipFree(b->d_obs));
    b->d_obs = nullptr;
    HIPCHK(hipMalloc((void**)&b->d_obs, need * sizeof(int32_t)));
    b->obs_rows_cap = max_rows;
  }
  BbxParams p; fill_params(b, &p);
  p.nsteps = 0; p.set_budget = 1; p.agent = BBX_AGENT_EXTERNAL; p.auto_reset = 0;
  p.obs = b->d_obs; p.obs_rows = max_rows; p.obs_fill = fill; p.trace = nullptr;
  int rc = launch(b, p, 0);
  if (rc) return rc;
  rc = finish(b, 0);
  if (rc) return rc;
  HIPCHK(hipMemcpy(out, b->d_obs, need * sizeof(int32_t), hipMemcpyDeviceToHost));
  return BBX_OK;
}

int bbx_cols(const bbx_batch* b) { return b ? 2 * b->nvars * b->k : 0; }
int bbx_nvars(const bbx_batch* b) { return b ? b->nvars : 0; }
int bbx_batch_size(const bbx_batch* b) { return b ? b->B : 0; }

}  // extern "C"
