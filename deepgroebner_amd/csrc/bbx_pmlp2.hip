// PMLP policy with TWO hidden layers on the observation block (ParallelMultilayerPerceptron(hidden_layers=[h1, h2]),
// networks.py:522-571: ParallelEmbeddingLayer :49-95 with two dense layers + ParallelDecidingLayer :414-460), evaluated and
// sampled on the device like the one-layer kernel of bbx_pmlp.h:
//     logit_r = w3 . relu(W2^T relu(W1^T x_r + b1) + b2) + b3,  log-softmax over the rows of an environment, inverse-CDF draw.
// fp32 throughout, both layers on the matrix cores (v_mfma_f32_16x16x4_f32: exact f32), one wavefront per environment,
// 16 rows per tile (pair sets average 19 rows: tiles of 32 rows computed 35 row slots per environment, tiles of 16 compute 27).
//
// D = A x B with A = weights [unit][k], B = activations [k][row]: lane l of a 16x16x4 tile supplies A[unit = l & 15][k = l >> 4]
// and B[k = l >> 4][row = l & 15] and receives D[unit = 4 (l >> 4) + v][row = l & 15] in register v = 0..3.  So after layer 1 a
// lane holds, for ITS row, units 16 blk + 4 (l >> 4) + v of every block blk — and layer 2 needs, as the B operand of k-step s,
// h1[row = l & 15][k(s, l >> 4)]: a sum over k may run in ANY order, so the k-steps are numbered the way the accumulators
// already lie,  k(s, g) = 16 (s >> 2) + 4 g + (s & 3),  and the B operand of k-step s IS register (s & 3) of block (s >> 2) of
// the same lane, after relu: no transpose, no LDS round trip, no cross-lane traffic between the layers.  The permutation is
// folded into the PREPARED second-layer weights (bbx_pmlp2_prepare), stored so that a lane fetches the A operands of four
// k-steps with one 16-byte LDS read:
//     A2[blk2][s4][lane][j] = W2[k(4 s4 + j, lane >> 4)][16 blk2 + (lane & 15)]
// staged into LDS once per workgroup (64 KB for 128 x 128) together with b2 and w3.  Two blocks of 16 second-layer units are
// in flight together: a dependent chain of this instruction issues every 40 cycles, two interleaved chains every 32.  The
// deciding layer is an in-lane dot over the accumulators and two exchanges between the four lane groups of a row.
//
// Workgroups of 8 waves, two per CU = four waves per SIMD at <= 128 registers: a wave spends a third of its life waiting
// (rows, the observation tile, the LDS reads of the next A operands, the softmax) and the matrix cores were busy 39 % of a
// launch with two waves per SIMD (SQ_VALU_MFMA_BUSY_CYCLES: 25 us of 64) — more waves are what fills them, not a better
// distribution of the environments: sorting them by tile count (even loads per wave) and handing them out through a ticket
// counter (dynamic) were both built; the first changed nothing, the second was slower (same-address atomics from eight XCDs).
#include "bbx_device.h"
#include "bbx_pmlp.h"

// prepared weights (floats): W1p [4 KS][HP1] | b1p [HP1] | A2 [HP2 / 16][HP1 / 16][64][4] | b2p [HP2] | w3p [HP2] | b3, pad
// HP = the layer padded to 64 or 128 units, KS = k-steps of four columns built in
__host__ __device__ constexpr int pmlp2_hp_for(int hidden) { return hidden <= 64 ? 64 : 128; }
__host__ __device__ constexpr int pmlp2_ks_for(int cols) { const int ks = (cols + 3) / 4; return ks <= 3 ? 3 : ks <= 8 ? 8 : 16; }
__host__ __device__ constexpr int pmlp2_prepared_floats(int cols, int h1, int h2) {
  return (4 * pmlp2_ks_for(cols) + 1) * pmlp2_hp_for(h1) + pmlp2_hp_for(h1) * pmlp2_hp_for(h2) + 2 * pmlp2_hp_for(h2) + 4;
}
constexpr int PMLP2_WAVES = 8;

__global__ void bbx_pmlp2_prepare_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                         const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                                         int cols, int h1, int h2, float* __restrict__ out) {
  const int HP1 = pmlp2_hp_for(h1), HP2 = pmlp2_hp_for(h2), K1 = 4 * pmlp2_ks_for(cols), S4 = HP1 / 16;
  const int o_b1 = K1 * HP1, o_a2 = o_b1 + HP1, o_b2 = o_a2 + HP1 * HP2, o_w3 = o_b2 + HP2, o_b3 = o_w3 + HP2, total = o_b3 + 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < o_b1) { const int k = i / HP1, h = i - k * HP1; v = (k < cols && h < h1) ? w1[(size_t)k * h1 + h] : 0.f; }
    else if (i < o_a2) { const int h = i - o_b1; v = h < h1 ? b1[h] : 0.f; }
    else if (i < o_b2) {
      const int t = i - o_a2, j = t & 3, lane = (t >> 2) & 63, q = t >> 8, s4 = q % S4, blk2 = q / S4;
      const int s = 4 * s4 + j;
      const int k = 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3), unit = 16 * blk2 + (lane & 15);
      v = (k < h1 && unit < h2) ? w2[(size_t)k * h2 + unit] : 0.f;
    }
    else if (i < o_w3) { const int h = i - o_b2; v = h < h2 ? b2[h] : 0.f; }
    else if (i < o_b3) { const int h = i - o_w3; v = h < h2 ? w3[h] : 0.f; }
    else v = i == o_b3 ? b3[0] : 0.f;
    out[i] = v;
  }
}

template <int HP1, int HP2, int KS>
__global__ __launch_bounds__(PMLP2_WAVES * WAVE, 2) void bbx_pmlp2_act_kernel(const int32_t* __restrict__ obs, const int32_t* __restrict__ rows, int B,
                                                                             int obs_rows, int cols, const float* __restrict__ wp,
                                                                             const float* __restrict__ u, int32_t* __restrict__ actions,
                                                                             float* __restrict__ logprobs, int lgcap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NK1 = HP1 / 16, NK2 = HP2 / 16, A2F = HP1 * HP2;   // blocks of 16 units per layer
  constexpr int HS = NK1 / 2;                                       // A operands are requested half a block pair at a time
  float* a2 = (float*)smem;
  const float* W1p = wp;
  const float* b1p = wp + 4 * KS * HP1;
  const float* a2g = b1p + HP1;
  // (A2 | b2p | w3p are contiguous in the prepared buffer: one copy; the biases and the deciding weights of a block of units
  // are then LDS reads next to its A operands, not trips to memory in front of every block's MFMAs)
  for (int i = (int)threadIdx.x; i < (A2F + 2 * HP2) / 4; i += (int)blockDim.x) ((bbx_f32x4*)a2)[i] = ((const bbx_f32x4*)a2g)[i];
  __syncthreads();
  const float* b2l = a2 + A2F;
  const float* w3l = b2l + HP2;
  const int lane = lane_id(), wave = uni((int)(threadIdx.x / WAVE)), nw = (int)blockDim.x / WAVE;
  float* lg = a2 + A2F + 2 * HP2 + (size_t)wave * lgcap;                      // logits of this wave's environment
  const int lr = lane & 15, lg4 = lane >> 4;
  const float b3 = a2g[A2F + 2 * HP2];
  // A wave's tiles run one after the other, so with one environment per wave a launch lasts as long as its LARGEST pair set
  // (B = 4096: the mean is 1.7 tiles, the maximum 5-7; measured 60 us per launch for 25 us of matrix-core time, the same with
  // two or four waves per SIMD).  The wave therefore takes only the first two tiles of its environment itself; further tiles
  // go to a queue of the workgroup and are shared out among its eight waves after a barrier (logits land in the owner's LDS
  // buffer), and the owner samples after a second barrier.
  float* lg_base = a2 + A2F + 2 * HP2;
  int* s_env = (int*)(lg_base + (size_t)nw * lgcap);                          // [nw] environment of each wave this round
  int* s_n = s_env + nw;                                                      // [nw] its row count
  int* s_q = s_n + nw;                                                        // queue length
  unsigned short* queue = (unsigned short*)(s_q + 1);                         // [nw * 64] (wave << 8) | tile
  for (int base = (int)blockIdx.x * nw; base < B; base += (int)gridDim.x * nw) {
    const int env = base + wave;
    const bool valid = env < B;
    int n = valid ? uni(rows[env]) : 0;
    const float uu = valid ? u[env] : 0.f;
    n = n < obs_rows ? n : obs_rows; n = n < PMLP_MAXROWS ? n : PMLP_MAXROWS; n = n > 0 ? n : 0;
    const int T = (n + 15) >> 4;
    __syncthreads();                                                          // (the previous round is over: logits, queue)
    if (lane == 0) { s_env[wave] = env; s_n[wave] = n; }
    if (threadIdx.x == 0) *s_q = 0;
    __syncthreads();
    if (lane < T - 2) queue[atomicAdd(s_q, 1)] = (unsigned short)((wave << 8) | (2 + lane));
  for (int phase = 0; phase < 2; phase++) {
    int cnt = T < 2 ? T : 2;
    if (phase == 1) { __syncthreads(); cnt = uni(*s_q); }
    for (int it = phase == 0 ? 0 : wave; it < cnt; it += (phase == 0 ? 1 : nw)) {
      int w = wave, t = it;
      if (phase == 1) { const int e = uni((int)queue[it]); w = e >> 8; t = e & 255; }
      const int tn = uni(s_n[w]), r0 = 16 * t;
      const int32_t* ob = obs + (size_t)uni(s_env[w]) * obs_rows * cols;
      float* lgt = lg_base + (size_t)w * lgcap;
    {
      int r = r0 + lr; r = r < obs_rows ? r : obs_rows - 1;                   // inside the block whatever the row count is
      const int32_t* xr = ob + (size_t)r * cols;
      float xa[KS];
#pragma unroll
      for (int s = 0; s < KS; s++) {
        const int k = 4 * s + lg4;
        const int32_t xi = xr[k < cols ? k : 0];
        xa[s] = k < cols ? (float)xi : 0.f;
      }
      // ---- layer 1: h[blk][v] = relu(b1 + sum_k W1[k][unit] x[row][k]), unit = 16 blk + 4 (lane >> 4) + v
      // (the first-layer weights stay in memory: L1 hits; the base pointer is opaque per tile so that the optimiser does not
      // hoist KS x NK1 loads out of the loops into registers the 128-register budget does not have)
      bbx_f32x4 h[NK1];
#pragma unroll
      for (int j = 0; j < NK1; j++) h[j] = *(const bbx_f32x4*)(b1p + 16 * j + 4 * lg4);
      const float* W1q = W1p + lg4 * HP1 + lr;
      asm volatile("" : "+v"(W1q));
#pragma unroll
      for (int s = 0; s < KS; s++) {
#pragma unroll
        for (int j = 0; j < NK1; j++) h[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(W1q[4 * s * HP1 + 16 * j], xa[s], h[j], 0, 0, 0);
        if (KS > 3 && (s & 1)) __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int j = 0; j < NK1; j++) {
        h[j].x = h[j].x > 0.f ? h[j].x : 0.f; h[j].y = h[j].y > 0.f ? h[j].y : 0.f;
        h[j].z = h[j].z > 0.f ? h[j].z : 0.f; h[j].w = h[j].w > 0.f ? h[j].w : 0.f;
      }
      // ---- layer 2 + deciding layer, two blocks of 16 units at a time
      float part = 0.f;
#pragma clang loop unroll(disable)
      for (int b2i = 0; b2i < NK2; b2i += 2) {
        bbx_f32x4 acc0 = *(const bbx_f32x4*)(b2l + 16 * b2i + 4 * lg4), acc1 = *(const bbx_f32x4*)(b2l + 16 * b2i + 16 + 4 * lg4);
        const bbx_f32x4 w0 = *(const bbx_f32x4*)(w3l + 16 * b2i + 4 * lg4), w1v = *(const bbx_f32x4*)(w3l + 16 * b2i + 16 + 4 * lg4);
        const bbx_f32x4* ap = (const bbx_f32x4*)a2 + (size_t)b2i * NK1 * 64 + lane;
#pragma unroll
        for (int hf = 0; hf < 2; hf++) {
          // the A operands of half of both blocks (2 HS reads of 16 bytes per lane) are requested before their MFMAs
          bbx_f32x4 av0[HS], av1[HS];
#pragma unroll
          for (int q = 0; q < HS; q++) { av0[q] = ap[(hf * HS + q) * 64]; av1[q] = ap[(NK1 + hf * HS + q) * 64]; }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < HS; q++) {
            const bbx_f32x4 hb = h[hf * HS + q];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[q].x, hb.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[q].x, hb.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[q].y, hb.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[q].y, hb.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[q].z, hb.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[q].z, hb.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[q].w, hb.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[q].w, hb.w, acc1, 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        part = fmaf(acc0.x > 0.f ? acc0.x : 0.f, w0.x, part); part = fmaf(acc0.y > 0.f ? acc0.y : 0.f, w0.y, part);
        part = fmaf(acc0.z > 0.f ? acc0.z : 0.f, w0.z, part); part = fmaf(acc0.w > 0.f ? acc0.w : 0.f, w0.w, part);
        part = fmaf(acc1.x > 0.f ? acc1.x : 0.f, w1v.x, part); part = fmaf(acc1.y > 0.f ? acc1.y : 0.f, w1v.y, part);
        part = fmaf(acc1.z > 0.f ? acc1.z : 0.f, w1v.z, part); part = fmaf(acc1.w > 0.f ? acc1.w : 0.f, w1v.w, part);
      }
      part += __shfl_xor(part, 16, WAVE);                                     // the other lane groups hold the row's other units
      part += __shfl_xor(part, 32, WAVE);
      if (lg4 == 0 && r0 + lr < tn) lgt[r0 + lr] = part + b3;
    }
    }
  }
    __syncthreads();                                                          // (every tile of the workgroup's environments is in)
    if (valid) {
      if (n <= 0) { if (lane == 0) { actions[env] = 0; logprobs[env] = 0.f; } }
      else pmlp_sample(lg, n, env, uu, actions, logprobs);
    }
  }
}

extern "C" int bbx_pmlp2_floats(int cols, int h1, int h2) { return pmlp2_prepared_floats(cols, h1, h2); }

extern "C" int bbx_launch_pmlp2_prepare(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                                        int cols, int h1, int h2, float* out, hipStream_t stream) {
  hipLaunchKernelGGL(bbx_pmlp2_prepare_kernel, dim3(64), dim3(256), 0, stream, w1, b1, w2, b2, w3, b3, cols, h1, h2, out);
  return (int)hipGetLastError();
}

extern "C" int bbx_launch_pmlp2_act(const int32_t* obs, const int32_t* rows, int B, int obs_rows, int cols, const float* wp, int h1, int h2,
                                    const float* u, int32_t* actions, float* logprobs, int max_blocks, hipStream_t stream) {
  const int waves = PMLP2_WAVES, hp1 = pmlp2_hp_for(h1), hp2 = pmlp2_hp_for(h2), ks = pmlp2_ks_for(cols);
  int lgcap = obs_rows < PMLP_MAXROWS ? obs_rows : PMLP_MAXROWS;              // logits per wave: what the block can hold
  lgcap = (lgcap + 63) / 64 * 64;
  const size_t ml = ((size_t)hp1 * hp2 + 2 * hp2) * sizeof(float) + (size_t)waves * lgcap * sizeof(float) + (size_t)(2 * waves + 1) * sizeof(int) + (size_t)waves * 64 * sizeof(unsigned short);
  int blocks = (B + waves - 1) / waves;
  blocks = blocks < max_blocks ? blocks : max_blocks;
#define BBX_P2(N1, N2, K) do { \
    static size_t set_ = 0;            /* (once per size: the call is not free) */ \
    if (set_ < ml) { \
      hipError_t err_ = hipFuncSetAttribute((const void*)bbx_pmlp2_act_kernel<N1, N2, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ml); \
      if (err_ != hipSuccess) return (int)err_; \
      set_ = ml; } \
    hipLaunchKernelGGL((bbx_pmlp2_act_kernel<N1, N2, K>), dim3(blocks), dim3(waves * WAVE), ml, stream, obs, rows, B, obs_rows, cols, wp, u, actions, logprobs, lgcap); } while (0)
#define BBX_P2_K(N1, N2) do { if (ks == 3) BBX_P2(N1, N2, 3); else if (ks == 8) BBX_P2(N1, N2, 8); else BBX_P2(N1, N2, 16); } while (0)
  if (hp1 == 64 && hp2 == 64) BBX_P2_K(64, 64); else if (hp1 == 64) BBX_P2_K(64, 128); else if (hp2 == 64) BBX_P2_K(128, 64); else BBX_P2_K(128, 128);
#undef BBX_P2_K
#undef BBX_P2
  return (int)hipGetLastError();
}
