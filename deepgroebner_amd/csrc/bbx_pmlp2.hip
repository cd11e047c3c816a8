// PMLP policy with TWO hidden layers on the observation block (ParallelMultilayerPerceptron(hidden_layers=[h1, h2]),
// networks.py:522-571: ParallelEmbeddingLayer :49-95 with two dense layers + ParallelDecidingLayer :414-460), evaluated and
// sampled on the device like the one-layer kernel of bbx_pmlp.h:
//     logit_r = w3 . relu(W2^T relu(W1^T x_r + b1) + b2) + b3,  log-softmax over the rows of an environment, inverse-CDF draw.
// fp32 throughout, both layers on the matrix cores (v_mfma_f32_32x32x2_f32), one wavefront per environment, 32 rows per tile.
//
// Layer 1 is the tile of bbx_pmlp.h with its accumulators kept: D1[unit][row], lane l holds row (l & 31) and, in register v
// of unit block nb, unit 32 nb + (v & 3) + 8 (v >> 2) + 4 (l >> 5).  Layer 2 needs, as the B operand of k-step s, the value
// h1[row = l & 31][k = k(s, l >> 5)] — and the sum over k may run in ANY order, so the k-steps are numbered the way the
// accumulators already lie:  k(s, half) = 32 (s >> 4) + (s & 3) + 8 ((s & 15) >> 2) + 4 half.  Then the B operand of k-step s
// IS accumulator register (s & 15) of unit block (s >> 4) of the same lane, after relu: no transpose, no LDS round trip, no
// cross-lane traffic between the layers.  The permutation is folded into the PREPARED second-layer weights
// (bbx_pmlp2_prepare), stored so that a lane fetches the A operands of four k-steps with one 16-byte LDS read:
//     A2[nb2][s4][lane][j] = W2[k(4 s4 + j, lane >> 5)][32 nb2 + (lane & 31)]
// staged into LDS once per workgroup (64 KB for 128 x 128); workgroups are persistent over the batch (environment = wave
// index + k * waves in the grid), so the staging is paid 2 x #CU times per launch, not per environment.
// The deciding layer is the in-lane dot of bbx_pmlp.h over the second layer's accumulators.
#include "bbx_device.h"
#include "bbx_pmlp.h"

// prepared weights (floats): W1p [2 KS][32 NB1] | b1p [32 NB1] | A2 [NB2][4 NB1][64][4] | b2p [32 NB2] | w3p [32 NB2] | b3, pad
__host__ __device__ constexpr int pmlp2_nb_for(int hidden) { return hidden <= 64 ? 2 : 4; }
__host__ __device__ constexpr int pmlp2_ks_for(int cols) { const int ks = (cols + 1) / 2; return ks <= 6 ? 6 : ks <= 16 ? 16 : 32; }
__host__ __device__ constexpr int pmlp2_a2_floats(int nb1, int nb2) { return 1024 * nb1 * nb2; }
__host__ __device__ constexpr int pmlp2_prepared_floats(int cols, int h1, int h2) {
  return (2 * pmlp2_ks_for(cols) + 1) * 32 * pmlp2_nb_for(h1) + pmlp2_a2_floats(pmlp2_nb_for(h1), pmlp2_nb_for(h2)) + 2 * 32 * pmlp2_nb_for(h2) + 4;
}

__global__ void bbx_pmlp2_prepare_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                         const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                                         int cols, int h1, int h2, float* __restrict__ out) {
  const int NB1 = pmlp2_nb_for(h1), NB2 = pmlp2_nb_for(h2), HP1 = 32 * NB1, HP2 = 32 * NB2, K2 = 2 * pmlp2_ks_for(cols);
  const int o_b1 = K2 * HP1, o_a2 = o_b1 + HP1, o_b2 = o_a2 + pmlp2_a2_floats(NB1, NB2), o_w3 = o_b2 + HP2, o_b3 = o_w3 + HP2, total = o_b3 + 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < o_b1) { const int k = i / HP1, h = i - k * HP1; v = (k < cols && h < h1) ? w1[(size_t)k * h1 + h] : 0.f; }
    else if (i < o_a2) { const int h = i - o_b1; v = h < h1 ? b1[h] : 0.f; }
    else if (i < o_b2) {
      const int t = i - o_a2, j = t & 3, lane = (t >> 2) & 63, q = t >> 8, s4 = q % (4 * NB1), nb2 = q / (4 * NB1);
      const int s = 4 * s4 + j, vv = s & 15;
      const int k = 32 * (s >> 4) + (vv & 3) + 8 * (vv >> 2) + 4 * (lane >> 5), unit = 32 * nb2 + (lane & 31);
      v = (k < h1 && unit < h2) ? w2[(size_t)k * h2 + unit] : 0.f;
    }
    else if (i < o_w3) { const int h = i - o_b2; v = h < h2 ? b2[h] : 0.f; }
    else if (i < o_b3) { const int h = i - o_w3; v = h < h2 ? w3[h] : 0.f; }
    else v = i == o_b3 ? b3[0] : 0.f;
    out[i] = v;
  }
}

template <int NB1, int NB2, int KS>
__global__ __launch_bounds__(256, 2) void bbx_pmlp2_act_kernel(const int32_t* __restrict__ obs, const int32_t* __restrict__ rows, int B, int obs_rows,
                                                               int cols, const float* __restrict__ wp, const float* __restrict__ u,
                                                               int32_t* __restrict__ actions, float* __restrict__ logprobs, int lgcap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HP1 = 32 * NB1, HP2 = 32 * NB2, A2F = pmlp2_a2_floats(NB1, NB2);
  float* a2 = (float*)smem;
  const float* W1p = wp;
  const float* b1p = wp + 2 * KS * HP1;
  const float* a2g = b1p + HP1;
  const float* b2p = a2g + A2F;
  const float* w3p = b2p + HP2;
  // (A2 | b2p | w3p are contiguous in the prepared buffer: one copy; the biases and the deciding weights of a block of units
  // are then LDS reads next to its A operands, not trips to memory in front of every block's MFMAs)
  for (int i = (int)threadIdx.x; i < (A2F + 2 * HP2) / 4; i += (int)blockDim.x) ((bbx_f32x4*)a2)[i] = ((const bbx_f32x4*)a2g)[i];
  __syncthreads();
  const float* b2l = a2 + A2F;
  const float* w3l = b2l + HP2;
  const int lane = lane_id(), wave = uni((int)(threadIdx.x / WAVE)), nw = (int)blockDim.x / WAVE;
  float* lg = a2 + A2F + 2 * HP2 + (size_t)wave * lgcap;                                // logits of this wave's environment
  const int lr = lane & 31, lk = lane >> 5;
  const float b3 = w3p[HP2];
  // (The launch ends with its unluckiest SIMD: an environment costs one tile per 32 rows — 1.11 tiles on average on
  // 3-20-10-weighted — and four environments share a SIMD: 33 us of matrix-core time become ~70 us per launch at B = 4096,
  // whether 256, 512 or 1024 workgroups share the batch.  Handing the environments out through a ticket counter was built and
  // measured slower, 88-125 us: same-address atomics from all eight XCDs cost more than the imbalance.  What is left to try is
  // a finer unit: tiles of 16 rows (v_mfma_f32_16x16x4_f32) or logits written per tile and sampled by a second kernel.)
  for (int env = (int)blockIdx.x * nw + wave; env < B; env += (int)gridDim.x * nw) {
    int n = uni(rows[env]);
    const float uu = u[env];
    n = n < obs_rows ? n : obs_rows; n = n < PMLP_MAXROWS ? n : PMLP_MAXROWS;
    if (n <= 0) { if (lane == 0) { actions[env] = 0; logprobs[env] = 0.f; } continue; }
    const int32_t* ob = obs + (size_t)env * obs_rows * cols;
    for (int r0 = 0; r0 < n; r0 += 32) {
      int r = r0 + lr; r = r < obs_rows ? r : obs_rows - 1;                   // inside the block whatever the row count is
      const int32_t* xr = ob + (size_t)r * cols;
      float xa[KS];
#pragma unroll
      for (int s2 = 0; s2 < KS; s2++) {
        const int k = 2 * s2 + lk;
        const int32_t xi = xr[k < cols ? k : 0];
        xa[s2] = k < cols ? (float)xi : 0.f;
      }
      // ---- layer 1: h[nb][v] = relu(b1 + sum_k W1[k][unit] x[row][k])
      bbx_f32x16 h[NB1];
#pragma unroll
      for (int j = 0; j < NB1; j++) {
        const int ub = j * 32 + 4 * lk;
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bbx_f32x4 bv = *(const bbx_f32x4*)(b1p + ub + 8 * q);
          h[j][4 * q] = bv.x; h[j][4 * q + 1] = bv.y; h[j][4 * q + 2] = bv.z; h[j][4 * q + 3] = bv.w;
        }
      }
#pragma unroll
      for (int s2 = 0; s2 < KS; s2++)
#pragma unroll
        for (int j = 0; j < NB1; j++) h[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(W1p[(2 * s2 + lk) * HP1 + j * 32 + lr], xa[s2], h[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < NB1; j++)
#pragma unroll
        for (int v = 0; v < 16; v++) h[j][v] = h[j][v] > 0.f ? h[j][v] : 0.f;
      // ---- layer 2 + deciding layer, one block of 32 units at a time
      float part = 0.f;
#pragma clang loop unroll(disable)
      for (int nb2 = 0; nb2 < NB2; nb2++) {
        const int ub = nb2 * 32 + 4 * lk;
        bbx_f32x16 acc;
        bbx_f32x4 wv[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bbx_f32x4 bv = *(const bbx_f32x4*)(b2l + ub + 8 * q);
          acc[4 * q] = bv.x; acc[4 * q + 1] = bv.y; acc[4 * q + 2] = bv.z; acc[4 * q + 3] = bv.w;
          wv[q] = *(const bbx_f32x4*)(w3l + ub + 8 * q);
        }
        // the A operands of the whole block (4 NB1 reads of 16 bytes per lane) are requested before the first of its MFMAs:
        // read just in time, every eighth MFMA waited a full LDS round trip with one or two waves per SIMD to cover it
        const bbx_f32x4* ap = (const bbx_f32x4*)a2 + (size_t)nb2 * (4 * NB1) * 64 + lane;
        bbx_f32x4 av[4 * NB1];
#pragma unroll
        for (int s4 = 0; s4 < 4 * NB1; s4++) av[s4] = ap[s4 * 64];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s4 = 0; s4 < 4 * NB1; s4++) {
          const bbx_f32x4 a = av[s4];
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, h[(4 * s4) >> 4][(4 * s4) & 15], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, h[(4 * s4 + 1) >> 4][(4 * s4 + 1) & 15], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, h[(4 * s4 + 2) >> 4][(4 * s4 + 2) & 15], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, h[(4 * s4 + 3) >> 4][(4 * s4 + 3) & 15], acc, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bbx_f32x4 w = wv[q];
          const float h0 = acc[4 * q], h1 = acc[4 * q + 1], h2 = acc[4 * q + 2], h3 = acc[4 * q + 3];
          part = fmaf(h0 > 0.f ? h0 : 0.f, w.x, part); part = fmaf(h1 > 0.f ? h1 : 0.f, w.y, part);
          part = fmaf(h2 > 0.f ? h2 : 0.f, w.z, part); part = fmaf(h3 > 0.f ? h3 : 0.f, w.w, part);
        }
      }
      const float logit = part + __shfl_xor(part, 32, WAVE) + b3;             // the other half of the row's units
      if (lk == 0 && r0 + lr < n) lg[r0 + lr] = logit;
    }
    pmlp_sample(lg, n, env, uu, actions, logprobs);
    wave_sync();                                                              // (the logits are rewritten for the next environment)
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
  const int waves = 4, nb1 = pmlp2_nb_for(h1), nb2 = pmlp2_nb_for(h2), ks = pmlp2_ks_for(cols);
  int lgcap = obs_rows < PMLP_MAXROWS ? obs_rows : PMLP_MAXROWS;              // logits per wave: what the block can hold
  lgcap = (lgcap + 63) / 64 * 64;
  const size_t ml = ((size_t)pmlp2_a2_floats(nb1, nb2) + 2 * 32 * nb2) * sizeof(float) + (size_t)waves * lgcap * sizeof(float);
  int blocks = (B + waves - 1) / waves;
  blocks = blocks < max_blocks ? blocks : max_blocks;
#define BBX_P2(N1, N2, K) do { \
    static size_t set_ = 0;            /* (once per size: the call is not free, and one in ~900 of them stalls for 40 ms) */ \
    if (set_ < ml) { \
      hipError_t err_ = hipFuncSetAttribute((const void*)bbx_pmlp2_act_kernel<N1, N2, K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ml); \
      if (err_ != hipSuccess) return (int)err_; \
      set_ = ml; } \
    hipLaunchKernelGGL((bbx_pmlp2_act_kernel<N1, N2, K>), dim3(blocks), dim3(waves * WAVE), ml, stream, obs, rows, B, obs_rows, cols, wp, u, actions, logprobs, lgcap); } while (0)
#define BBX_P2_K(N1, N2) do { if (ks == 6) BBX_P2(N1, N2, 6); else if (ks == 16) BBX_P2(N1, N2, 16); else BBX_P2(N1, N2, 32); } while (0)
  if (nb1 == 2 && nb2 == 2) BBX_P2_K(2, 2); else if (nb1 == 2) BBX_P2_K(2, 4); else if (nb2 == 2) BBX_P2_K(4, 2); else BBX_P2_K(4, 4);
#undef BBX_P2_K
#undef BBX_P2
  return (int)hipGetLastError();
}
