// PMLP policy on the observation block (included by the binomial / fast / aux translation units: the step kernels of
// the binomial classes have the policy built in).
#pragma once

// ------------------------------------------------------------------ the consumer of the observation block: PMLP policy
// The reference's default policy (ParallelMultilayerPerceptron, networks.py:522-571 = ParallelEmbeddingLayer :49-95 with
// one dense layer + ParallelDecidingLayer :414-460) evaluated on the padded observation block and sampled, without the
// block leaving the device:   logit_r = w2 . relu(W1^T x_r + b1) + b2   over the |P| rows of an environment,
// log-softmax over them, and one action drawn by inverse CDF from a caller-supplied uniform number u (so that the
// torch module of deepgroebner_amd/rollout.py reproduces the draw).  fp32 throughout.  One wavefront per environment.
// Padded rows (beyond rows[e]) never reach a logit: the -1 padding the reference masks out (networks.py:94-95, 456-457)
// plays no part.
constexpr int PMLP_MAXROWS = BBX_POLICY_MAX_ROWS;
// logits per wave in LDS: what the caller's block can hold, at most PMLP_MAXROWS
// (no block, obs_rows <= 0 — a policy rollout that keeps no observations: every row the kernels score)
__host__ __device__ constexpr int pmlp_lgcap(int obs_rows) { return obs_rows > 0 && obs_rows < PMLP_MAXROWS ? obs_rows : PMLP_MAXROWS; }
__device__ __forceinline__ float wave_sum_f32(float x) {    // sum over the 64 lanes, valid in lane 63
#define BBX_DPPADD(ctrl, rmask) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xF, false));
  BBX_DPPADD(0x111, 0xF) BBX_DPPADD(0x112, 0xF) BBX_DPPADD(0x114, 0xF) BBX_DPPADD(0x118, 0xF) BBX_DPPADD(0x142, 0xA) BBX_DPPADD(0x143, 0xC)
#undef BBX_DPPADD
  return x;
}
// inclusive prefix sum over the lanes (every lane; the wave total in lane 63) and the wave maximum (lane 63): the DPP
// doubling sequence — lanes without a source keep `old`, the identity of the operation
__device__ __forceinline__ float wave_max_f32(float x) {
#define BBX_DPPMAX(ctrl, rmask) { const float t_ = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, x), __builtin_bit_cast(int, x), ctrl, rmask, 0xF, false)); x = t_ > x ? t_ : x; }
  BBX_DPPMAX(0x111, 0xF) BBX_DPPMAX(0x112, 0xF) BBX_DPPMAX(0x114, 0xF) BBX_DPPMAX(0x118, 0xF) BBX_DPPMAX(0x142, 0xA) BBX_DPPMAX(0x143, 0xC)
#undef BBX_DPPMAX
  return x;
}
__device__ __forceinline__ float lane63_f32(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 63)); }
// log-softmax over the n logits of one environment (in LDS) and the inverse-CDF draw; lane 0 writes action and
// log-probability; returns the sampled row (wave-uniform)
__device__ __forceinline__ int pmlp_sample(const float* lg, int n, int env, float uu, int32_t* __restrict__ actions, float* __restrict__ logprobs) {
  const int lane = lane_id();
  wave_sync();
  float mx = -3.0e38f;
  for (int r = lane; r < n; r += WAVE) { const float t = lg[r]; mx = t > mx ? t : mx; }
  mx = lane63_f32(wave_max_f32(mx));
  float se = 0.f;
  for (int r = lane; r < n; r += WAVE) se += __expf(lg[r] - mx);
  se = lane63_f32(wave_sum_f32(se));
  const float logz = mx + __logf(se);
  // inverse CDF over the rows in order: the first row whose cumulative probability exceeds u (the last one on round-off)
  const float target = uu * se;
  float run = 0.f; int pick = -1;
  for (int base = 0; base < n; base += WAVE) {
    const int r = base + lane;
    const float e = r < n ? __expf(lg[r] - mx) : 0.f;
    const float c = wave_sum_f32(e);                         // inclusive prefix within the wave
    const uint64_t hit = ballot64(r < n && run + c > target);
    if (hit) { pick = base + (int)__builtin_ctzll(hit); break; }
    run += lane63_f32(c);
  }
  if (pick < 0) pick = n - 1;
  if (lane == 0) { actions[env] = pick; logprobs[env] = lg[pick] - logz; }
  return uni(pick);
}

// The hidden layer on the matrix cores, exact f32 (v_mfma_f32_32x32x2_f32 = an fmaf chain), everything in registers: a
// tile is 32 hidden units x 32 rows of the block, D[unit][row] = sum_k W1[k][unit] x[row][k] + b1[unit]:
//   A operand  lane l: W1[2s + (l >> 5)][unit = 32 nb + (l & 31)]        (coalesced loads, the same for every wave: L1)
//   B operand  lane l: x[row = r0 + (l & 31)][2s + (l >> 5)]             (KS loads per lane)
//   C / D      lane l, register v: unit 32 nb + (v & 3) + 8 (v >> 2) + 4 (l >> 5), row r0 + (l & 31); starts at b1[unit]
// so that a ROW's hidden vector lies along the registers of the two lanes l and l + 32: relu and the dot with w2 are
// in-lane multiply-adds over the accumulator registers and one exchange between the wave's halves finishes the logit —
// no cross-lane reduction tree, no LDS staging, no barrier.  The weights come PREPARED (bbx_pmlp_prepare: zero-padded to
// 2 KS x 32 NB, so every load is unconditional at a compile-time offset), and nothing the first tile needs depends on
// another load: row count, uniform number, rows (read without knowing the count: the block holds obs_rows rows whatever
// it means) and weights are all requested before the first wait — at four waves per SIMD a dependent trip to memory
// costs more than the arithmetic of a tile.
// KS = k-steps built in (>= ceil(cols / 2)), NB = unit blocks (>= ceil(hidden / 32), a power of two).
typedef float bbx_f32x16 __attribute__((ext_vector_type(16)));
typedef float bbx_f32x4 __attribute__((ext_vector_type(4)));
__host__ __device__ constexpr int pmlp_ks_for(int cols) {   // the built-in k-step counts
  const int ks = (cols + 1) / 2;
  return ks <= 3 ? 3 : ks <= 6 ? 6 : ks <= 10 ? 10 : ks <= 16 ? 16 : 32;
}
__host__ __device__ constexpr int pmlp_nb_for(int hidden) { const int nb = (hidden + 31) / 32; return nb <= 1 ? 1 : nb <= 2 ? 2 : nb <= 4 ? 4 : 8; }
// prepared weights (floats): W1p [2 KS][32 NB] | b1p [32 NB] | w2p [32 NB] | b2 | pad to a multiple of 4
__host__ __device__ constexpr int pmlp_prepared_floats(int cols, int hidden) {
  return (2 * pmlp_ks_for(cols) + 2) * 32 * pmlp_nb_for(hidden) + 4;
}
// one tile: the logit (without b2) of row (lane & 31) from the lane's B operands xa[]; G unit blocks in flight together
// (registers: 32 G + G KS)
// (WP: where the prepared weights live — `const float*` in memory, or an LDS pointer when a rollout kernel has staged them
// once per launch)
template <int NB, int KS, int G, class WP = const float*>
__device__ __forceinline__ float pmlp_tile(const float (&xa)[KS], WP wp, int lr, int lk) {
  constexpr int HP = 32 * NB;
  const WP b1p = wp + 2 * KS * HP;
  const WP w2p = b1p + HP;
  const WP wl = wp + lk * HP + lr;                          // my A-operand column: + 2 s HP + 32 nb
  float part = 0.f;
#pragma clang loop unroll(disable)
  for (int g0 = 0; g0 < NB; g0 += G) {
    bbx_f32x16 acc[G];
    bbx_f32x4 wv[G][4];
    float wa[G][KS];
#pragma unroll
    for (int j = 0; j < G; j++) {
      const int ub = (g0 + j) * 32 + 4 * lk;                 // + (v & 3) + 8 (v >> 2): the units of my accumulator registers
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const WP bp = b1p + ub + 8 * q, vp = w2p + ub + 8 * q;
        acc[j][4 * q] = bp[0]; acc[j][4 * q + 1] = bp[1]; acc[j][4 * q + 2] = bp[2]; acc[j][4 * q + 3] = bp[3];   // (one 16-byte load)
        bbx_f32x4 w4; w4.x = vp[0]; w4.y = vp[1]; w4.z = vp[2]; w4.w = vp[3];
        wv[j][q] = w4;
      }
#pragma unroll
      for (int s2 = 0; s2 < KS; s2++) wa[j][s2] = wl[2 * s2 * HP + (g0 + j) * 32];
    }
#pragma unroll
    for (int s2 = 0; s2 < KS; s2++)
#pragma unroll
      for (int j = 0; j < G; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[j][s2], xa[s2], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < G; j++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const bbx_f32x4 w = wv[j][q];
        const float h0 = acc[j][4 * q], h1 = acc[j][4 * q + 1], h2 = acc[j][4 * q + 2], h3 = acc[j][4 * q + 3];
        part = fmaf(h0 > 0.f ? h0 : 0.f, w.x, part); part = fmaf(h1 > 0.f ? h1 : 0.f, w.y, part);
        part = fmaf(h2 > 0.f ? h2 : 0.f, w.z, part); part = fmaf(h3 > 0.f ? h3 : 0.f, w.w, part);
      }
  }
  return part + __shfl_xor(part, 32, WAVE);                  // the other half of the row's units
}
template <int NB, int KS>
__device__ __forceinline__ int pmlp_act_wave(char* smem, int env, bool live, const int32_t* __restrict__ obs, const int32_t* __restrict__ rows,
                                             int obs_rows, int cols, const float* __restrict__ wp, const float* __restrict__ u,
                                             int32_t* __restrict__ actions, float* __restrict__ logprobs) {
  constexpr int HP = 32 * NB;
  constexpr int G = (KS <= 10 ? 2 : 1) < NB ? (KS <= 10 ? 2 : 1) : NB;   // unit blocks in flight together
  const int lane = lane_id(), wave = uni((int)(threadIdx.x / WAVE));
  float* lg = (float*)smem + (size_t)wave * pmlp_lgcap(obs_rows);   // logits of this wave's environment
  if (!live) return 0;
  const int lr = lane & 31, lk = lane >> 5;
  const int nraw = rows[env];
  const float uu = u[env];
  const float b2 = wp[(size_t)(2 * KS + 2) * HP];
  const int32_t* ob = obs + (size_t)env * obs_rows * cols;
  int n = 0;
  for (int r0 = 0;; r0 += 32) {
    int r = r0 + lr; r = r < obs_rows ? r : obs_rows - 1;    // inside the block whatever the row count is
    const int32_t* xr = ob + (size_t)r * cols;
    float xa[KS];
#pragma unroll
    for (int s2 = 0; s2 < KS; s2++) {
      const int k = 2 * s2 + lk;
      const int32_t xi = xr[k < cols ? k : 0];
      xa[s2] = k < cols ? (float)xi : 0.f;
    }
    const float logit = pmlp_tile<NB, KS, G>(xa, wp, lr, lk);
    if (r0 == 0) { n = uni(nraw); n = n < obs_rows ? n : obs_rows; n = n < PMLP_MAXROWS ? n : PMLP_MAXROWS; }
    if (lk == 0 && r0 + lr < n) lg[r0 + lr] = logit + b2;
    if (r0 + 32 >= n) break;
  }
  if (n <= 0) { if (lane == 0) { actions[env] = 0; logprobs[env] = 0.f; } return 0; }
  wave_sync();
  return pmlp_sample(lg, n, env, uu, actions, logprobs);
}
template <int NB, int KS>
__global__ __launch_bounds__(256, 4) void bbx_pmlp_act_mfma_kernel(const int32_t* __restrict__ obs, const int32_t* __restrict__ rows, int B, int obs_rows,
                                                                   int cols, const float* __restrict__ wp, const float* __restrict__ u,
                                                                   int32_t* __restrict__ actions, float* __restrict__ logprobs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int env = blockIdx.x * ((int)blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
  pmlp_act_wave<NB, KS>(smem, env, env < B, obs, rows, obs_rows, cols, wp, u, actions, logprobs);
}
// LDS bytes of pmlp_act_wave for a workgroup of `waves` waves (the logits)
__host__ __device__ constexpr size_t pmlp_lds_bytes(int waves, int obs_rows) { return (size_t)waves * pmlp_lgcap(obs_rows) * sizeof(float); }
