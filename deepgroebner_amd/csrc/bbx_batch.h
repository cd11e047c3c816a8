// The batch handle of libbbx's C ABI and the internals its translation units share (bbx_api.cpp: creation, kernels of a
// launch, waits, stepping; bbx_api_session.cpp: what a call becomes — persistent / mailbox sessions, recorded steps;
// bbx_api_value.cpp: value(); bbx_api_state.cpp: introspection, generators, text format).
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <random>
#include <string>
#include <deque>
#include <vector>

#include "bbx_host.h"
#include "bbx_ideals.h"

extern "C" int bbx_launch_step(const BbxParams* p, int kind, int envs_per_block, hipStream_t stream);
extern "C" int bbx_launch_clone(const char* src_recs, char* dst_recs, const BbxLayout* L, const int32_t* src, const int32_t* dst, int n,
                                const uint32_t* seeds, int keep_counters, int seed_std, int ngen, uint8_t* flags, hipStream_t stream);
extern "C" int bbx_launch_value_resort(char* recs, const BbxLayout* L, int n, const uint8_t* flags, hipStream_t stream);
extern "C" int bbx_launch_value_collect(const char* recs, uint32_t rec_bytes, int n, double* out2, hipStream_t stream);
extern "C" int bbx_launch_relayout(const char* src_recs, char* dst_recs, const BbxLayout* Ls, const BbxLayout* Ld, int B, hipStream_t stream);
extern "C" int bbx_launch_ctl(unsigned long long* ctl, unsigned long long value, hipStream_t stream);
extern "C" int bbx_launch_gather_lite(const char* recs, uint32_t rec_bytes, int B, void* out, hipStream_t stream);
extern "C" int bbx_launch_gather_hdr(const char* recs, uint32_t rec_bytes, int B, BbxHdr* out, hipStream_t stream);
extern "C" int bbx_launch_scatter_queue(const uint32_t* stage, int n, uint32_t ring_words, uint32_t* q, int32_t* tail, hipStream_t stream);
extern "C" int bbx_launch_obs_pack(const int32_t* padded, int cap, int cols, const int32_t* rows, int B, int32_t* off, int32_t* packed, hipStream_t stream);
extern "C" int bbx_launch_init(char* recs, uint32_t rec_bytes, int B, const uint32_t* agent_seeds, hipStream_t stream);
extern "C" int bbx_launch_mark_reset(char* recs, uint32_t rec_bytes, int B, const uint8_t* mask, hipStream_t stream);
extern "C" int bbx_launch_pmlp_prepare(const float* w1, const float* b1, const float* w2, float b2, int cols, int hidden, float* out, hipStream_t stream);
extern "C" int bbx_launch_pmlp_act(const int32_t* obs, const int32_t* rows, int B, int obs_rows, int cols, const float* wp, int hidden, const float* u,
                                   int32_t* actions, float* logprobs, hipStream_t stream);


struct bbx_gen {
  std::unique_ptr<bbx::IdealGen> g;
  bbx::HIdeal last;
};

struct bbx_batch {
  int B = 0, device = 0, k = 1, nvars = 0, W = 2;
  int elim = 0, rewards = 0, sort_input = 0, sort_reducers = 1;
  bool fixed = false, binom = false, listed = false;   // listed: the ideals come from a caller's list (bbx_create_ideals)
  BbxLayout L{}, LL{};
  uint16_t* d_inv = nullptr;           // GF(32003) inverse table: one per device and process (bbx_host::inv_table), never freed
  std::vector<std::unique_ptr<bbx::IdealGen>> gens;   // one per environment (one shared when fixed)
  uint32_t slot_words = 0, nslots = 0;
  std::vector<uint32_t> h_q;          // host mirror of the ideal queue
  std::vector<int32_t> h_tail, h_head;
  std::vector<BbxHdr> h_hdr;
  std::vector<int32_t> h_lite;        // per environment {status, q_head, budget, nP}: what is polled after every launch
  bool q_dirty = true;
  std::vector<uint8_t> q_dirty_env;
  // ideals drawn on the device (binomial distributions): the table the kernels read, the per-environment engine state
  // lives in the record headers (BbxHdr.gen_rng); the host-side generators and the ideal queue are then unused
  uint32_t* d_gen = nullptr; size_t gen_words = 0; bool device_gen = false;
  std::shared_ptr<uint32_t> gen_owner;   // the table is immutable: copies of a handle share it (d_gen == gen_owner.get())
  std::vector<std::string> gen_error;   // per environment: a generator failure met while drawing ahead (see fill_queues)
  // device
  char* d_recs = nullptr;
  uint32_t* d_q = nullptr;
  int32_t* d_tail = nullptr;
  // one device block polled after every launch: lite[B][4] {status, q_head, budget, |P|} | rewards f64[B] | rows i32[B] |
  // dones u8[B]; the kernels write it themselves, the host fetches it with ONE copy into pinned memory
  char* d_out = nullptr; int32_t* d_lite = nullptr; double* d_rewards = nullptr; int32_t* d_rows = nullptr; uint8_t* d_dones = nullptr;
  char* h_io = nullptr; size_t io_bytes = 0;      // pinned mirror of d_out
  int32_t* h_act = nullptr;                       // pinned staging of host actions
  // small batches (the single-environment drop-in): the kernels read the actions from and write their outputs and the
  // observation straight into pinned host memory — no copy calls on the latency path, one stream synchronisation per step
  bool zero_copy = false, zc_active = false;
  bool poll_active = false; int poll_seq = 0, poll_misses = 0; unsigned polled_launches = 0;                   // the launch in flight signals completion through h_io (done_seq)
  char* zc_io_dev = nullptr; int32_t* zc_act_dev = nullptr;     // device-side addresses of h_io / h_act
  int32_t* h_zobs = nullptr; int32_t* zc_obs_dev = nullptr; size_t zobs_rows_cap = 0;
  // ragged observations (bbx_step_obs): device offsets [B+1] + packed rows, and their pinned mirror handed to the caller
  int32_t* d_obs_off = nullptr; int32_t* d_obs_packed = nullptr; int32_t* h_obs = nullptr; size_t obs_packed_cap = 0;
  uint32_t* h_stage = nullptr; uint32_t* d_stage = nullptr; size_t stage_words = 0;   // queue refill staging (pinned / device)
  int32_t* d_actions = nullptr; uint8_t* d_mask = nullptr; uint32_t* d_seeds = nullptr;
  int32_t* d_obs = nullptr; size_t obs_rows_cap = 0;
  BbxTraceRec* d_trace = nullptr; int trace_cap = 0;
  BbxHdr* d_hdr = nullptr;            // compact header copy (bbx_gather_hdr_kernel)
  // scratch for value(): cloned records, their headers, source indices, agent seeds, results
  char* d_vrecs = nullptr; BbxHdr* d_vhdr = nullptr; int32_t* d_vsrc = nullptr; uint32_t* d_vseeds = nullptr; double* d_vvals = nullptr;
  int vcap = 0;
  // HIP-event timing of the step-kernel launches (bbx_timing)
  bool accounting = true;             // count algorithmic bytes (bbx_accounting)
  bool timing = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_open;
  double kernel_ms = 0.0; int kernel_launches = 0;
  std::vector<char> h_out;
  // the rollout in flight (so bbx_sync can finish environments that waited for ideals)
  BbxParams last{};
  BbxParams cap_last{}; bool cap_valid = false, cap_obs_external = false, cap_policy_rollout = false;   // the last call recorded into a HIP graph (bbx_graph_replayed)
  int async_chain = 0;                // asynchronous steps with caller-supplied actions queued since the last wait (finish())
  bool cap_stale = false; std::vector<void*> retired;   // the records were enlarged after the recording: the old arrays stay allocated (replays write there)
  hipStream_t last_stream = 0;
  bool in_flight = false;
  bool policy_rollout = false;           // the launch in flight is a policy rollout (bbx_policy_rollout_device)
  int staged = 0, fast = 0, envs_per_block = 4;
  int fast_G = 0, fast_P = 0;          // capacities of the register/LDS-resident class (BbxParams::fast_G)
  int wide = 0;                       // > 0: waves per environment of the wide (one workgroup per environment) class
  int wide_terms = 0;                 // forced LDS capacity of the wide class (caps.wide_lds_terms), 0 = automatic
  int32_t* d_wide_done = nullptr;     // wide class: workgroups that have left the launch's first kernel (BbxParams::wide_tail)
  int ncu = 0;                        // compute units of the device
  bool device_async = false;          // the launch in flight came through a *_device entry point (no host poll per step)
  bool obs_external = false;          // the launch in flight writes observations into a caller-owned block: rows cut for
                                      // lack of space are an error the caller must hear about (bbx_sync)
  // persistent sessions (bbx_persistent): see BbxParams::ctl
  bool ps_enabled = false, ps_active = false;
  // host mailbox sessions (BbxParams::mbox): host-driven steps of small zero-copy batches on the register/LDS-resident class
  bool ps_mbox = false;                 // the session in progress is one
  unsigned long long* h_mbox = nullptr; // its control word, in pinned host memory, and the device's address of it
  unsigned long long* mbox_dev = nullptr;
  unsigned long long api_epoch = 0, mbox_epoch = ~0ull;   // launches / waits on the handle so far; the count as the last host step left it
  int mbox_streak = 0;                  // host steps in a row with nothing else on the handle in between (a loop: worth a session)
  int mbox_misses = 0;                  // steps whose result did not arrive through the mailbox in time (three in a row: no more mailbox sessions)
  unsigned long long* d_ctl = nullptr;
  hipStream_t ps_stream = nullptr, ps_ctl_stream = nullptr;   // the session's kernel / the writes to its control word
  hipEvent_t ps_ev = nullptr;
  long long ps_target = 0;            // steps issued since the session began
  std::deque<std::pair<double, long long>> ps_recent;   // (host time in ms, steps) of the latest calls: how much may still be owed when the session closes
  BbxParams ps_p{};                   // the parameters of the call that began it (later calls must match to join)
  BbxPolicy ps_pol{};                 // ... and its policy arguments (ps_p.policy points here), when it is a session of policy steps
  int ps_sessions = 0, ps_joined = 0, ps_kernels = 0; // statistics: sessions begun, calls that joined a running one, kernels
  int32_t* d_clone_idx = nullptr; int clone_cap = 0;   // bbx_clone_envs: source / destination indices on the device
  std::mt19937_64 value_rng;          // seeds of value("random") / value("sample") rollouts when the caller gives none
  bool gen_to_wide = false;           // general class with <= 16-byte monomials: long-polynomial environments continue in the wide class
  bool no_growth = false;             // bbx_caps.no_growth: the configured capacities are hard limits (BBX_E_CAPACITY)
  int grow_events = 0;                // times the records were enlarged (bbx_capacities)
  long long step_kernels = 0;         // kernels enqueue() has launched for this handle (bbx_kernels_launched: what a call costs in launches)
  bbx_batch() = default;
  bbx_batch(const bbx_batch&) = delete;
  bbx_batch& operator=(const bbx_batch&) = delete;
  ~bbx_batch();                       // frees every device / pinned allocation (also on half-built handles)
};


namespace bbx_host {
// wait for the launch in flight, serve environments that need the host (queued ideals, larger records, the kernels of a
// session), surface errors; the handle is left with nothing in flight
int finish(bbx_batch* b, hipStream_t stream);
int read_headers(bbx_batch* b, hipStream_t stream = 0);
void fill_params(bbx_batch* b, BbxParams* p);
int grow_records(bbx_batch* b, unsigned need, int env, hipStream_t stream);
const char* status_name(int s);
// bbx_api.cpp
int fill_queues(bbx_batch* b, int min_avail = 1, hipStream_t stream = 0);
int enqueue(bbx_batch* b, const BbxParams& p0, bool resume, hipStream_t stream);   // the kernels of one logical launch
// bbx_api_session.cpp
int launch(bbx_batch* b, BbxParams& p, hipStream_t stream, bool obs_external = false, bool device_async = false);
int ps_write_ctl(bbx_batch* b, bool stop);
int session_kernel(bbx_batch* b, bool first, hipStream_t after, bool sliced, bool behind_after = false);   // behind_after: ordered behind what `after` (possibly the NULL stream) holds
int session_close(bbx_batch* b, bool wait, hipStream_t then, bool sliced);
bool session_same_call(const BbxParams& a, const BbxParams& c);
bool mbox_eligible(const bbx_batch* b);
int mbox_step(bbx_batch* b, BbxParams& p, bool* used);
}  // namespace bbx_host
