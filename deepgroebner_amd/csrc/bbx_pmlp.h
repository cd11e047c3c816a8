// PMLP policy on the observation block (included by bbx_kernels.hip before the step kernels: the fast class has a
// policy + step launch).
#pragma once

// ------------------------------------------------------------------ the consumer of the observation block: PMLP policy
// The reference's default policy (ParallelMultilayerPerceptron, networks.py:522-571 = ParallelEmbeddingLayer :49-95 with
// one dense layer + ParallelDecidingLayer :414-460) evaluated on the padded observation block and sampled, without the
// block leaving the device:   logit_r = w2 . relu(W1^T x_r + b1) + b2   over the |P| rows of an environment,
// log-softmax over them, and one action drawn by inverse CDF from a caller-supplied uniform number u (so that the
// torch module of deepgroebner_amd/rollout.py reproduces the draw).  fp32 throughout.
// One wavefront per environment.  Environments have ~20 rows but the layer has 128+ units, so the lanes are dealt over
// the HIDDEN UNITS (lane l owns units l, l + 64, ... with their weights in registers: no LDS, no weight traffic in the
// loop); a row of the block is the same for every lane — wave-uniform loads through the scalar cache — and its logit is
// a DPP sum over the lanes.  Padded rows (beyond rows[e]) are never touched: the -1 padding the reference masks out
// (networks.py:94-95, 456-457) simply is not read.
constexpr int PMLP_MAXROWS = 1024;
__device__ __forceinline__ float wave_sum_f32(float x) {    // sum over the 64 lanes, valid in lane 63
#define BBX_DPPADD(ctrl, rmask) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xF, false));
  BBX_DPPADD(0x111, 0xF) BBX_DPPADD(0x112, 0xF) BBX_DPPADD(0x114, 0xF) BBX_DPPADD(0x118, 0xF) BBX_DPPADD(0x142, 0xA) BBX_DPPADD(0x143, 0xC)
#undef BBX_DPPADD
  return x;
}
// log-softmax over the n logits of one environment (in LDS) and the inverse-CDF draw; lane 0 writes action and log-probability
__device__ __forceinline__ int pmlp_sample(const float* lg, int n, int env, const float* __restrict__ u, int32_t* __restrict__ actions,
                                            float* __restrict__ logprobs) {
  const int lane = lane_id();
  wave_sync();
  float mx = -3.0e38f;
  for (int r = lane; r < n; r += WAVE) { const float t = lg[r]; mx = t > mx ? t : mx; }
  for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(mx, o, WAVE); mx = t > mx ? t : mx; }
  float se = 0.f;
  for (int r = lane; r < n; r += WAVE) se += __expf(lg[r] - mx);
  for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o, WAVE);
  const float logz = mx + __logf(se);
  // inverse CDF over the rows in order: the first row whose cumulative probability exceeds u (the last one on round-off)
  const float target = u[env] * se;
  float run = 0.f; int pick = -1;
  for (int base = 0; base < n; base += WAVE) {
    const int r = base + lane;
    const float e = r < n ? __expf(lg[r] - mx) : 0.f;
    float c = e;                                             // inclusive prefix within the wave
    for (int o = 1; o < WAVE; o <<= 1) { const float t = __shfl_up(c, o, WAVE); if (lane >= o) c += t; }
    const uint64_t hit = ballot64(r < n && run + c > target);
    if (hit) { pick = base + (int)__builtin_ctzll(hit); break; }
    run += __shfl(c, WAVE - 1, WAVE);
  }
  if (pick < 0) pick = n - 1;
  if (lane == 0) { actions[env] = pick; logprobs[env] = lg[pick] - logz; }
  return uni(pick);
}

// The hidden layer on the matrix cores, exact f32 (v_mfma_f32_32x32x2_f32 = an fmaf chain): a tile is 32 rows of the
// block x 32 hidden units, K = the row's columns two at a time.  A operand: the rows, staged through LDS as floats (lane
// l: row l & 31, column 2s + (l >> 5)); B operand = W1 from LDS; the accumulators start at b1.  Then relu, the dot with
// w2 in-lane over the NB unit blocks, and a 32-lane DPP sum per accumulator register: lane 31 / lane 63 end up with the
// logits of rows (v & 3) + 8 (v >> 2) (+ 4 for the upper half) of the tile.  Per environment this costs about as many
// instructions as the vector kernel below spends on sampling alone; what remains is launch and memory latency.
typedef float bbx_f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float half_sum_f32(float x) {    // sums over lanes 0..31 and 32..63, valid in lanes 31 and 63
#define BBX_DPPADD(ctrl, rmask) x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, rmask, 0xF, false));
  BBX_DPPADD(0x111, 0xF) BBX_DPPADD(0x112, 0xF) BBX_DPPADD(0x114, 0xF) BBX_DPPADD(0x118, 0xF) BBX_DPPADD(0x142, 0xA)
#undef BBX_DPPADD
  return x;
}
template <int NB>                                          // hidden units / 32 (padded)
__device__ __forceinline__ int pmlp_act_wave(char* smem, int env, bool live, const int32_t* __restrict__ obs, const int32_t* __restrict__ rows,
                                             int obs_rows, int cols, const float* __restrict__ w1, const float* __restrict__ b1,
                                             const float* __restrict__ w2, float b2, int hidden, const float* __restrict__ u,
                                             int32_t* __restrict__ actions, float* __restrict__ logprobs) {
  // called by every thread of the workgroup (one barrier inside); returns the sampled row of the wave's environment
  constexpr int HP = 32 * NB;
  const int lane = lane_id(), wave = uni((int)(threadIdx.x / WAVE));
  const int ks = (cols + 1) / 2;                            // k-steps
  const int sp = 2 * ks + 1;                                // LDS row stride of the staged rows (odd: conflict-free column reads)
  float* w1s = (float*)smem;                                // [2 ks][HP], zero padded
  float* lg = w1s + (size_t)2 * ks * HP + (size_t)wave * (PMLP_MAXROWS + WAVE * sp);
  float* xs = lg + PMLP_MAXROWS;                            // 64 rows of the block at a time, as floats
  for (int i = (int)threadIdx.x; i < 2 * ks * HP; i += (int)blockDim.x) {
    const int f = i / HP, h = i - f * HP;
    w1s[i] = (f < cols && h < hidden) ? w1[(size_t)f * hidden + h] : 0.f;
  }
  __syncthreads();
  if (!live) return 0;
  const int lr = lane & 31, lk = lane >> 5;
  float bj[NB], vj[NB];
#pragma unroll
  for (int nb = 0; nb < NB; nb++) { const int h = nb * 32 + lr; bj[nb] = h < hidden ? b1[h] : 0.f; vj[nb] = h < hidden ? w2[h] : 0.f; }
  int n = uni(rows[env]); n = n < obs_rows ? n : obs_rows; n = n < PMLP_MAXROWS ? n : PMLP_MAXROWS;
  if (n <= 0) { if (lane == 0) { actions[env] = 0; logprobs[env] = 0.f; } return 0; }
  const int32_t* ob = obs + (size_t)env * obs_rows * cols;
  const int dq = WAVE / cols, dr = WAVE - dq * cols;        // lane stride 64 as (rows, columns)
  for (int c0 = 0; c0 < n; c0 += WAVE) {
    const int nr = n - c0 < WAVE ? n - c0 : WAVE;
    // the chunk's rows come in with coalesced loads, all in flight together; a zero column pads odd widths
    {
      int rr = lane / cols, f = lane - rr * cols;
      for (int i = lane; i < nr * cols; i += WAVE) {
        xs[rr * sp + f] = (float)ob[(size_t)c0 * cols + i];
        f += dr; rr += dq; if (f >= cols) { f -= cols; rr++; }
      }
      if (2 * ks != cols) for (int rr2 = lane; rr2 < nr; rr2 += WAVE) xs[rr2 * sp + cols] = 0.f;
    }
    wave_sync();
    for (int r0 = 0; r0 < nr; r0 += 32) {
      const int r = r0 + lr < nr ? r0 + lr : nr - 1;         // (the tail tile repeats the last row; its logits are not stored)
      const float* xr = xs + r * sp + lk;
      bbx_f32x16 acc[NB];
#pragma unroll
      for (int nb = 0; nb < NB; nb++)
#pragma unroll
        for (int v = 0; v < 16; v++) acc[nb][v] = bj[nb];
      const float* wr = w1s + (size_t)lk * HP + lr;
      for (int s2 = 0; s2 < ks; s2++) {
        const float a = xr[2 * s2];
#pragma unroll
        for (int nb = 0; nb < NB; nb++) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, wr[(size_t)2 * s2 * HP + nb * 32], acc[nb], 0, 0, 0);
      }
#pragma unroll
      for (int v = 0; v < 16; v++) {
        float pj = 0.f;
#pragma unroll
        for (int nb = 0; nb < NB; nb++) { const float hv = acc[nb][v]; pj = fmaf(hv > 0.f ? hv : 0.f, vj[nb], pj); }
        pj = half_sum_f32(pj);
        const int rr = r0 + (v & 3) + 8 * (v >> 2) + 4 * lk;
        if (lr == 31 && rr < nr) lg[c0 + rr] = pj + b2;
      }
    }
    wave_sync();
  }
  wave_sync();
  return pmlp_sample(lg, n, env, u, actions, logprobs);
}
template <int NB>
__global__ __launch_bounds__(256, 4) void bbx_pmlp_act_mfma_kernel(const int32_t* __restrict__ obs, const int32_t* __restrict__ rows, int B, int obs_rows,
                                                                   int cols, const float* __restrict__ w1, const float* __restrict__ b1,
                                                                   const float* __restrict__ w2, float b2, int hidden, const float* __restrict__ u,
                                                                   int32_t* __restrict__ actions, float* __restrict__ logprobs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int env = blockIdx.x * ((int)blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
  pmlp_act_wave<NB>(smem, env, env < B, obs, rows, obs_rows, cols, w1, b1, w2, b2, hidden, u, actions, logprobs);
}
// LDS bytes of pmlp_act_wave for a workgroup of `waves` waves
__host__ __device__ constexpr size_t pmlp_lds_bytes(int nbp, int cols, int waves) {
  return ((size_t)2 * ((cols + 1) / 2) * 32 * nbp + (size_t)waves * (PMLP_MAXROWS + WAVE * (2 * ((cols + 1) / 2) + 1))) * sizeof(float);
}

template <int CP4, int UPL>                                // padded columns / 4; hidden units per lane
__global__ __launch_bounds__(256) void bbx_pmlp_act_kernel(const int32_t* __restrict__ obs, const int32_t* __restrict__ rows, int B, int obs_rows,
                                                           int cols, const float* __restrict__ w1, const float* __restrict__ b1,
                                                           const float* __restrict__ w2, float b2, int hidden, const float* __restrict__ u,
                                                           int32_t* __restrict__ actions, float* __restrict__ logprobs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int cp = 4 * CP4;
  const int lane = lane_id(), wave = uni((int)(threadIdx.x / WAVE));
  const int env = blockIdx.x * (blockDim.x / WAVE) + wave;
  if (env >= B) return;
  float* lg = (float*)smem + (size_t)wave * (PMLP_MAXROWS + WAVE * cp);   // logits of this wave's environment
  float* xs = lg + PMLP_MAXROWS;                             // 64 rows of the block at a time, as floats, rows padded to cp
  // my hidden units: weights in registers
  float wj[UPL][cp], bj[UPL], vj[UPL];
#pragma unroll
  for (int q = 0; q < UPL; q++) {
    const int h = lane + q * WAVE;
    const bool on = h < hidden;
    bj[q] = on ? b1[h] : 0.f; vj[q] = on ? w2[h] : 0.f;
#pragma unroll
    for (int f = 0; f < cp; f++) wj[q][f] = (on && f < cols) ? w1[(size_t)f * hidden + h] : 0.f;
  }
  int n = uni(rows[env]); n = n < obs_rows ? n : obs_rows; n = n < PMLP_MAXROWS ? n : PMLP_MAXROWS;
  if (n <= 0) { if (lane == 0) { actions[env] = 0; logprobs[env] = 0.f; } return; }
  const int32_t* ob = obs + (size_t)env * obs_rows * cols;
  for (int r0 = 0; r0 < n; r0 += WAVE) {
    const int nr = n - r0 < WAVE ? n - r0 : WAVE;
    // the chunk's rows come in with coalesced loads (all in flight together) and are read back as wave-uniform
    // 16-byte LDS reads: the row loop itself never waits on memory
    for (int i = lane; i < nr * cols; i += WAVE) { const int rr = i / cols, f = i - rr * cols; xs[rr * cp + f] = (float)ob[(size_t)r0 * cols + i]; }
    if (cp != cols) for (int i = lane; i < nr * (cp - cols); i += WAVE) { const int rr = i / (cp - cols), f = cols + i - rr * (cp - cols); xs[rr * cp + f] = 0.f; }
    wave_sync();
    for (int rb = 0; rb < nr; rb += 4) {                    // four rows at a time: four independent FMA chains and DPP sums in flight
      float part[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int r = rb + j < nr ? rb + j : nr - 1;         // (the chunk's tail repeats its last row)
        float x[cp];
#pragma unroll
        for (int f4 = 0; f4 < CP4; f4++) { const float4 v = *(const float4*)(xs + r * cp + 4 * f4); x[4 * f4] = v.x; x[4 * f4 + 1] = v.y; x[4 * f4 + 2] = v.z; x[4 * f4 + 3] = v.w; }
        float pj = 0.f;
#pragma unroll
        for (int q = 0; q < UPL; q++) {
          float acc = bj[q];
#pragma unroll
          for (int f = 0; f < cp; f++) acc = fmaf(x[f], wj[q][f], acc);
          pj = fmaf(acc > 0.f ? acc : 0.f, vj[q], pj);
        }
        part[j] = pj;
      }
#pragma unroll
      for (int j = 0; j < 4; j++) part[j] = wave_sum_f32(part[j]);
      if (lane == 63) {
#pragma unroll
        for (int j = 0; j < 4; j++) if (rb + j < nr) lg[r0 + rb + j] = part[j] + b2;
      }
    }
    wave_sync();
  }
  pmlp_sample(lg, n, env, u, actions, logprobs);
}
