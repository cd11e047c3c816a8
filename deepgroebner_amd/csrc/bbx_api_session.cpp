// libbbx.so — host side: persistent sessions (bbx_persistent), host mailbox sessions (4.1.4), recorded
// steps (HIP graphs), and launch(): the one place that decides which of them a call becomes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "../../include/bbx.h"
#include "bbx_common.h"
#include "bbx_ideals.h"
#include "bbx_batch.h"

namespace bbx_host {

// ---- persistent sessions (bbx_persistent; see BbxParams::ctl) --------------------------------------------------------
// A session runs as a sequence of kernels of at most PS_SLICE_TICKS each: a kernel whose slice is over leaves with what
// its environments still owe (BBX_ST_TIMESLICE) and the next call on the handle starts the next one.  The slice ends are
// where environments that left the register/LDS class (and were served by the HBM-resident pass meanwhile) come back
// to it — long kernels were slower per step because such stragglers stayed behind until the kernel ended (DESIGN.md
// 4.1.1) — and they bound how long a consumer that went away keeps the device busy.
constexpr uint32_t PS_SLICE_TICKS = 1000000u;            // 10 ms of the 100 MHz clock

int ps_write_ctl(bbx_batch* b, bool stop) {              // all writes to the control word travel on one stream, in order
  if (b->ps_mbox) {                                        // (a mailbox session's word is in host memory: the host writes it itself)
    std::atomic_thread_fence(std::memory_order_release);
    // (one environment: the step's action rides in bits 33.. of the word, + 1 — 0: look in the action buffer)
    const unsigned long long act = (b->B == 1 && !stop && b->h_act[0] >= 0) ? ((unsigned long long)(uint32_t)(b->h_act[0] + 1) << 33) : 0ull;
    __atomic_store_n(b->h_mbox, (unsigned long long)b->ps_target | (stop ? (1ull << 32) : 0ull) | act, __ATOMIC_RELEASE);
    return BBX_OK;
  }
  int lrc = bbx_launch_ctl(b->d_ctl, (unsigned long long)b->ps_target | (stop ? (1ull << 32) : 0ull), b->ps_ctl_stream);
  if (lrc) return fail(BBX_E_DEVICE, "control launch failed: %s", hipGetErrorString((hipError_t)lrc));
  return BBX_OK;
}

// Queue a kernel of the session on its stream: the first one (every environment starts with the steps issued so far) or a
// later one (every environment takes what it still owes of the total), behind the last write to the control word and
// behind what the caller queued on `after` (or null).  `sliced`: it leaves when its time slice is over.
int session_kernel(bbx_batch* b, bool first, hipStream_t after, bool sliced, bool behind_after) {
  if (!b->ps_mbox) {                                       // (a mailbox session's control word is written by the host, not on a stream)
    HIPCHK(hipEventRecord(b->ps_ev, b->ps_ctl_stream));
    HIPCHK(hipStreamWaitEvent(b->ps_stream, b->ps_ev, 0));
  }
  if (behind_after) {                                      // (`after` may be the NULL stream: the session stream does not wait for it by itself)
    HIPCHK(hipEventRecord(b->ps_ev, after));
    HIPCHK(hipStreamWaitEvent(b->ps_stream, b->ps_ev, 0));
  }
  BbxParams q = b->ps_p;
  q.recs = b->d_recs; q.L = b->L; q.ctl = b->ps_mbox ? b->mbox_dev : b->d_ctl; q.ctl_stats = b->d_ctl ? b->d_ctl + 8 : nullptr;
  q.mbox = b->ps_mbox ? 1 : 0;
  q.nsteps = (int32_t)b->ps_target; q.slice_ticks = sliced ? PS_SLICE_TICKS : 0u;
  q.set_budget = first ? 1 : 0; q.sess_target = first ? 0 : (int32_t)b->ps_target; q.pass = 0;
  b->ps_kernels++;
  return enqueue(b, q, false, b->ps_stream);
}

// End the session (asynchronously): the waves are told to stop once they have taken every step issued, and one more kernel
// of the session is queued behind the running one for whatever environments still owe — those whose wave had left before
// the last steps were issued (slice over, 20 ms without news) or had handed its environment to the HBM-resident class.
// `wait`: the stream `then` is made to wait for all of it.  `sliced`: the caller (finish) looks after kernels that
// leave at the end of their slice; otherwise the closing kernel runs to completion whatever it takes.
double ps_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void ps_note_steps(bbx_batch* b, long long nsteps) {
  const double t = ps_now_ms();
  b->ps_recent.emplace_back(t, nsteps);
  while (!b->ps_recent.empty() && b->ps_recent.front().first < t - 50.0) b->ps_recent.pop_front();
}

int session_close(bbx_batch* b, bool wait, hipStream_t then, bool sliced) {
  if (!b->ps_active) return BBX_OK;
  b->ps_active = false;
  int rc = ps_write_ctl(b, true);
  if (rc) return rc;
  const bool was_mbox = b->ps_mbox;
  if (!sliced) {
    // The host will not be there to start the next kernel when a slice ends, and ONE kernel without time limit would keep
    // every environment that leaves the register/LDS class on the HBM-resident pass until the very end (a straggler of
    // thousands of steps: measured 65 ms instead of 22 for 8 x 1024 steps).  So the rest is queued as a chain of sliced
    // kernels — as many as the steps issued lately can need at 5 us a step, at most 32; one that finds nothing owed costs
    // a few microseconds — and the kernel without limit behind them takes whatever is left after that.
    const double t = ps_now_ms();
    long long lately = 0;
    for (const auto& e : b->ps_recent) if (e.first >= t - 50.0) lately += e.second;
    long long chain = (lately + 2047) / 2048;
    chain = chain < 1 ? 1 : (chain > 32 ? 32 : chain);
    for (long long i = 0; i < chain; i++) { rc = session_kernel(b, false, nullptr, true); if (rc) return rc; }
  }
  b->ps_recent.clear();
  rc = session_kernel(b, false, nullptr, sliced);
  if (rc) return rc;
  b->last = b->ps_p; b->last.recs = b->d_recs; b->last.L = b->L;   // (a resumed pass continues from the budgets left in the headers)
  b->last.ctl = nullptr; b->last.sess_target = 0; b->last.set_budget = 0; b->last.policy = nullptr; b->last.mbox = 0;
  b->last_stream = b->ps_stream;
  if (was_mbox) { b->ps_mbox = false; b->zc_active = true; }   // (its outputs are in the pinned block: finish() reads them there)
  if (wait) {                                              // (`then` may be the null stream)
    HIPCHK(hipEventRecord(b->ps_ev, b->ps_stream));
    HIPCHK(hipStreamWaitEvent(then, b->ps_ev, 0));
  }
  return BBX_OK;
}

bool session_same_call(const BbxParams& a, const BbxParams& c) {
  return a.agent == c.agent && a.auto_reset == c.auto_reset && a.rewards == c.rewards && a.dones == c.dones && a.rows == c.rows &&
         a.obs == c.obs && a.obs_rows == c.obs_rows && a.obs_fill == c.obs_fill && a.obs_every_step == c.obs_every_step &&
         a.actions == c.actions && a.recs == c.recs;
}

// ---- launch: what a call becomes -------------------------------------------------------------------------------------
// A call that steps environments is one of three things, decided here and nowhere else:
//   L_CAPTURE  recorded into a HIP graph (the caller's stream is capturing): a pure function of device state, nothing for the
//              host to upload, no events, no second stream — bbx_graph_replayed restores it as the call in flight later;
//   L_SESSION  a call of a persistent session: the first (begins the session's first kernel) or a later one of the same shape
//              (raises the step total the running kernel looks at); a call of another shape ends the session first;
//   L_PLAIN    its own kernel(s) on the caller's stream.
enum LaunchKind { L_PLAIN, L_CAPTURE, L_SESSION_BEGIN, L_SESSION_JOIN };

static bool is_policy_step_call(const BbxParams& p) { return p.policy && p.policy->rollout == 1 && p.policy->post_obs; }   // bbx_policy_step_device

// asynchronous rollouts with a built-in agent (or per-step calls of the one-layer policy) on the register/LDS-resident class,
// lean and untraced, ideals drawn on the device (nothing for the host to do between launches), every wave resident at once
static bool session_admits(const bbx_batch* b, const BbxParams& p, bool device_async) {
  return b->ps_enabled && device_async && b->fast && b->staged && b->device_gen && !b->accounting && p.nsteps >= 1 && p.auto_reset &&
         (is_policy_step_call(p) || (!p.policy && p.obs_fill == 0 && (p.agent == BBX_AGENT_HASH || p.agent == BBX_AGENT_DEGREE || p.agent == BBX_AGENT_FIRST))) &&
         !(b->d_trace && b->trace_cap >= 1) && !b->timing && b->B <= 4096 && p.set_budget == 1;
}
static bool session_same_policy(const bbx_batch* b, const BbxParams& p) {
  const BbxPolicy* a = b->ps_p.policy; const BbxPolicy* c = p.policy;
  return (!a && !c) || (a && c && is_policy_step_call(p) && a->wp == c->wp && a->hidden == c->hidden && a->actions == c->actions &&
                        a->logprobs == c->logprobs && a->rewards_t == c->rewards_t && a->dones_t == c->dones_t && a->rows_t == c->rows_t &&
                        c->u == a->u + (size_t)b->ps_target * (size_t)b->B);
}

static int launch_session_join(bbx_batch* b, const BbxParams& p, hipStream_t stream, bool obs_external) {
  b->ps_target += p.nsteps;                                // the waves see the new total the next time they look
  ps_note_steps(b, p.nsteps);
  int rc = ps_write_ctl(b, false);
  if (rc) return rc;
  // the session's kernel may have left meanwhile (its slice was over, or no news for 20 ms): the next one
  if (hipStreamQuery(b->ps_stream) == hipSuccess) { rc = session_kernel(b, false, stream, true, true); if (rc) return rc; }
  b->ps_joined++;
  b->in_flight = true; b->obs_external = obs_external; b->device_async = true;
  return BBX_OK;
}
static int launch_session_begin(bbx_batch* b, const BbxParams& p, hipStream_t stream, bool obs_external) {
  b->ps_p = p; b->ps_p.ctl = nullptr;
  if (p.policy) { b->ps_pol = *p.policy; b->ps_p.policy = &b->ps_pol; }
  b->ps_target = p.nsteps; b->ps_active = true; b->ps_sessions++;
  b->ps_recent.clear(); ps_note_steps(b, p.nsteps);
  // The control word is the handle's, not the session's: the kernels that close the PREVIOUS session (queued on the session
  // stream, possibly not yet run) still poll it, and a new total written now would be theirs to take — steps of this session
  // under the last one's agent and buffers (found by scripts/fuzz_sessions.py: two calls of different shapes back to back).
  // The write therefore waits for everything queued on the session stream so far.
  if (!b->ps_mbox) {
    HIPCHK(hipEventRecord(b->ps_ev, b->ps_stream));
    HIPCHK(hipStreamWaitEvent(b->ps_ctl_stream, b->ps_ev, 0));
  }
  int rc = ps_write_ctl(b, false);
  if (rc) return rc;
  b->last = p; b->last.ctl = nullptr; b->last.policy = nullptr;
  b->policy_rollout = false; b->last_stream = b->ps_stream; b->in_flight = true; b->obs_external = obs_external; b->device_async = true;
  return session_kernel(b, true, stream, true, true);
}
static int launch_plain(bbx_batch* b, const BbxParams& p, hipStream_t stream, bool obs_external, bool device_async, bool capturing) {
  if (device_async && p.agent == BBX_AGENT_EXTERNAL && p.nsteps >= 1) b->async_chain++;
  b->last = p; b->last.ctl = nullptr;
  b->policy_rollout = p.policy && p.policy->rollout;
  b->last.policy = nullptr;                 // (a host pointer of the caller's frame: never kept)
  b->last_stream = stream;
  b->in_flight = true;
  b->obs_external = obs_external;
  b->device_async = device_async;
  if (capturing) {                          // what bbx_graph_replayed restores: the call the replays repeat
    b->cap_last = b->last; b->cap_valid = true; b->cap_stale = false; b->cap_obs_external = obs_external; b->cap_policy_rollout = b->policy_rollout;
  }
  return enqueue(b, p, false, stream);
}

int launch(bbx_batch* b, BbxParams& p, hipStream_t stream, bool obs_external, bool device_async) {
  b->api_epoch++;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (stream != nullptr && hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
  const bool capturing = cap != hipStreamCaptureStatusNone;
  if (capturing) {
    if (!device_async) return fail(BBX_E_UNSUPPORTED, "only the asynchronous device calls can be captured into a graph");
    if (!(b->device_gen || (b->fixed && !b->q_dirty))) return fail(BBX_E_UNSUPPORTED, "a batch whose ideals are drawn on the host cannot be captured into a graph");
    if (b->ps_enabled || b->ps_active || b->timing) return fail(BBX_E_UNSUPPORTED, "persistent sessions and kernel timing cannot be captured into a graph");
  }
  int rc = fill_queues(b, 1, stream);
  if (rc) return rc;
  const bool admits = !capturing && session_admits(b, p, device_async);
  LaunchKind kind = capturing ? L_CAPTURE : L_PLAIN;
  if (b->ps_active) {
    if (admits && !b->ps_mbox && session_same_policy(b, p) && session_same_call(b->ps_p, p) && b->ps_target + p.nsteps < (1ll << 30)) kind = L_SESSION_JOIN;
    else {                                  // something else: the session ends; what follows is ordered behind it
      rc = session_close(b, !admits, stream, false);
      if (rc) return rc;
      if (admits) kind = L_SESSION_BEGIN;
    }
  } else if (admits) kind = L_SESSION_BEGIN;
  switch (kind) {
    case L_SESSION_JOIN: return launch_session_join(b, p, stream, obs_external);
    case L_SESSION_BEGIN: return launch_session_begin(b, p, stream, obs_external);
    case L_CAPTURE: return launch_plain(b, p, stream, obs_external, device_async, true);
    default: return launch_plain(b, p, stream, obs_external, device_async, false);
  }
}

// ---- host mailbox sessions ---------------------------------------------------------------------------------------------
// A host-driven step of a small batch (the reference's usage: ONE environment stepped from Python, wrapped.pyx:23-26) used
// to be one kernel launch per step: ~14 us in the library for ~3 us of work.  On the register/LDS-resident class with
// device-drawn ideals the step calls of a loop feed ONE resident kernel instead (a persistent session, DESIGN.md 4.1.1, whose
// control word and action buffer are pinned host memory the host writes itself): a step is a store of the actions, a store of
// the control word, and a spin on the status words the kernel publishes with every step's outputs.  Everything else on
// the handle closes the session through the usual path (finish): it is never observable except in time.
bool mbox_eligible(const bbx_batch* b) {
  // (up to 8 environments: measured at 16 / 32 / 64 a session is no faster than — 24 / 34 / 61 against 24 / 28 / 32 us — the
  // zero-copy launch per step, whose one kernel serves all of them at once)
  return b->zero_copy && b->B <= 8 && b->fast && b->staged && b->device_gen && !b->accounting && !b->timing && !(b->d_trace && b->trace_cap >= 1) &&
         b->mbox_misses < 3 && !getenv("BBX_NO_MAILBOX");
}
// p: the step's parameters (external agent, zero-copy outputs, nsteps = 1).  Returns BBX_OK with the step taken and its outputs
// in the pinned block, or an error; *used = false: not taken here (the caller launches as before).
int mbox_step(bbx_batch* b, BbxParams& p, bool* used) {
  *used = false;
  int rc;
  const bool join = b->ps_active && b->ps_mbox && session_same_call(b->ps_p, p) && b->ps_target < (1ll << 30);
  if (!join) {
    // a session pays when the steps come in a row (a loop); a caller that does something else on the handle between steps
    // (value() per step, pg.py:461-465) is served by one launch per step as before: four steps in a row start a session
    b->mbox_streak = (b->api_epoch == b->mbox_epoch + 1) ? b->mbox_streak + 1 : 0;   // (+ 1: this call's own entry)
    if (b->mbox_streak < 4) return BBX_OK;
    if (!b->h_mbox) {                                       // (the session's streams exist from the first session on, not before)
      HIPCHK(hipHostMalloc((void**)&b->h_mbox, 64, hipHostMallocCoherent | hipHostMallocMapped));
      HIPCHK(hipHostGetDevicePointer((void**)&b->mbox_dev, b->h_mbox, 0));
      if (!b->ps_stream) {
        HIPCHK(hipStreamCreateWithFlags(&b->ps_stream, hipStreamNonBlocking));
        HIPCHK(hipStreamCreateWithFlags(&b->ps_ctl_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&b->ps_ev, hipEventDisableTiming));
      }
    }
    if (b->in_flight) { rc = finish(b, b->last_stream); if (rc) return rc; }
    rc = fill_queues(b, 1, nullptr);
    if (rc) return rc;
    for (int e = 0; e < b->B; e++) ((volatile int32_t*)b->h_io)[(size_t)e * 4] = 0;
    p.mbox = 1;
    b->ps_p = p; b->ps_p.ctl = nullptr; b->ps_p.policy = nullptr;
    b->ps_mbox = true; b->ps_active = true; b->ps_target = 1; b->ps_sessions++;
    b->ps_recent.clear();
    rc = ps_write_ctl(b, false);
    if (rc) return rc;
    b->last = p; b->last.ctl = nullptr; b->last.policy = nullptr; b->last.mbox = 0;
    b->policy_rollout = false; b->last_stream = b->ps_stream; b->in_flight = true; b->obs_external = false; b->device_async = false;
    rc = session_kernel(b, true, nullptr, true);
    if (rc) return rc;
  } else {
    b->ps_target++;
    rc = ps_write_ctl(b, false);
    if (rc) return rc;
    b->ps_joined++;
    if (hipStreamQuery(b->ps_stream) == hipSuccess) { rc = session_kernel(b, false, nullptr, true); if (rc) return rc; }   // (slice over, or idle for 20 ms)
  }
  // the step's sequence number on every environment's status word — or something else to look at
  const uint32_t want = (uint32_t)(b->ps_target % 16000) + 1u;
  const volatile int32_t* w = (const volatile int32_t*)b->h_io;
  const auto t0 = std::chrono::steady_clock::now();
  bool all = false, trouble = false, timed_out = false;
  for (unsigned spins = 0;; spins++) {
    all = true;
    for (int e = 0; e < b->B; e++) {
      const uint32_t v = (uint32_t)w[(size_t)e * 4];
      if ((v >> 17) != want) all = false;
      else if ((v & 0xffffu) != BBX_ST_OK || (v & BBX_LITE_OBS_TRUNC)) trouble = true;
    }
    if (all || trouble) break;
    if ((spins & 63) == 63) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(3)) { timed_out = true; break; }
      // the kernel may have left (its slice, 20 ms without news, an environment that needs the host): the next one takes the step
      if (hipStreamQuery(b->ps_stream) == hipSuccess) {
        bool seen = true;
        for (int e = 0; e < b->B; e++) seen = seen && (((uint32_t)w[(size_t)e * 4]) >> 17) == want;
        if (seen) { all = true; break; }
        bool stopped = false;                                // an environment that left with something to report ends the mailbox
        for (int e = 0; e < b->B; e++) { const uint32_t st = (uint32_t)w[(size_t)e * 4] & 0xffffu; stopped = stopped || (st != BBX_ST_OK && st != BBX_ST_TIMESLICE); }
        if (stopped) break;
        rc = session_kernel(b, false, nullptr, true);
        if (rc) return rc;
      }
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  *used = true;
  if (all && !trouble) {
    b->mbox_misses = 0;
    b->h_lite.resize((size_t)b->B * 4);
    memcpy(b->h_lite.data(), b->h_io, (size_t)b->B * 16);
    return BBX_OK;
  }
  // not through the mailbox (an error status, rows beyond the caller's block, an environment that left the class, no answer):
  // the session closes the usual way — its closing kernels take whatever step is still owed — and reports what there is
  if (timed_out) b->mbox_misses++;
  return finish(b, b->ps_stream);
}

}  // namespace bbx_host
using namespace bbx_host;

extern "C" {

int bbx_persistent(bbx_batch* b, int enable) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
  if (enable && !b->d_ctl) {
    HIPCHK(hipMalloc((void**)&b->d_ctl, 65536));          // (word 0: control, word 8: statistics; the rest: scripts/patches)
    HIPCHK(hipMemset(b->d_ctl, 0, 65536));
    HIPCHK(hipStreamCreateWithFlags(&b->ps_stream, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&b->ps_ctl_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&b->ps_ev, hipEventDisableTiming));
    HIPCHK(hipDeviceSynchronize());
  }
  b->ps_enabled = enable != 0;
  return BBX_OK;
}

int bbx_session_stats(bbx_batch* b, int64_t* out5) {   // out: 5 values
  if (!b || !out5) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  out5[0] = b->ps_sessions; out5[1] = b->ps_joined; out5[2] = 0; out5[3] = b->ps_kernels; out5[4] = 0;
  if (b->d_ctl) {
    if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
    unsigned long long v[2] = {0, 0};
    HIPCHK(hipMemcpy(v, b->d_ctl + 8, sizeof v, hipMemcpyDeviceToHost));
    out5[2] = (int64_t)v[0]; out5[4] = (int64_t)v[1];
  }
  return BBX_OK;
}

int bbx_kernels_launched(bbx_batch* b, int64_t* out) {
  if (!b || !out) return fail(BBX_E_ARG, "null argument");
  *out = b->step_kernels;
  return BBX_OK;
}

int bbx_join(bbx_batch* b, void* stream) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  HIPCHK(hipSetDevice(b->device)); b->api_epoch++;
  if (!b->ps_active) return BBX_OK;
  return session_close(b, true, (hipStream_t)stream, false);
}

int bbx_graph_replayed(bbx_batch* b, void* stream) {
  if (!b) return fail(BBX_E_ARG, "null argument");
  if (b->ps_active) return fail(BBX_E_UNSUPPORTED, "a persistent session is running on this handle");
  if (!b->cap_valid) return fail(BBX_E_ARG, "no asynchronous step or rollout of this handle has been recorded into a graph");
  HIPCHK(hipSetDevice(b->device));                        // (finish() below may enlarge the records: allocations go to the current device)
  if (b->cap_stale) {
    b->cap_valid = false; b->cap_stale = false;
    return fail(BBX_E_CAPACITY, "the records of this batch were enlarged after the step was recorded: the graph steps the retired copy, "
                                "the steps replayed since did not reach the batch — record the step again");
  }
  if (b->in_flight) { int rc = finish(b, b->last_stream); if (rc) return rc; }
  b->last = b->cap_last; b->policy_rollout = b->cap_policy_rollout; b->obs_external = b->cap_obs_external; b->device_async = true;
  b->in_flight = true; b->last_stream = (hipStream_t)stream;
  b->async_chain = 2;                                   // (any number of replays)
  return BBX_OK;
}

}  // extern "C"
