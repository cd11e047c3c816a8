// PMLP policy with TWO or THREE hidden layers on the observation block (ParallelMultilayerPerceptron(hidden_layers=[h1, h2]) or
// [h1, hm, h2]; the text below describes two: a middle layer is the second layer's code with its output kept in registers,
// which is again the layout the next layer wants — pmlp2_hidden<.., LAST = false>),
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
#include <atomic>
#include "bbx_device.h"
#include "bbx_pmlp.h"

// prepared weights (floats): W1p [4 KS][HP1] | b1p [HP1] | [AM [HPM / 16][HP1 / 16][64][4]] | A2 [HP2 / 16][HPI / 16][64][4] | [bMp [HPM]] |
// b2p [HP2] | wdp [HP2] | bd, pad          (bracketed: the optional middle hidden layer; HPI = HPM if there is one, else HP1)
// HP = the layer padded to 64 or 128 units, KS = k-steps of four columns built in
__host__ __device__ constexpr int pmlp2_hp_for(int hidden) { return hidden <= 64 ? 64 : 128; }
__host__ __device__ constexpr int pmlp2_ks_for(int cols) { const int ks = (cols + 3) / 4; return ks <= 3 ? 3 : ks <= 8 ? 8 : 16; }
__host__ __device__ constexpr int pmlp2_prepared_floats(int cols, int hp1, int hpm, int hp2) {   // (padded sizes; hpm = 0: two hidden layers)
  return (4 * pmlp2_ks_for(cols) + 1) * hp1 + hp1 * hpm + (hpm ? hpm : hp1) * hp2 + hpm + 2 * hp2 + 4;
}
constexpr int PMLP2_WAVES = 8;

// one permuted second-/third-layer matrix: element t of A[HPO / 16][HPI / 16][64][4]
__device__ __forceinline__ float pmlp2_perm(const float* __restrict__ w, int hi, int ho, int HPI, int t) {
  const int S4 = HPI / 16, j = t & 3, lane = (t >> 2) & 63, q = t >> 8, s4 = q % S4, blk = q / S4;
  const int s = 4 * s4 + j;
  const int k = 16 * (s >> 2) + 4 * (lane >> 4) + (s & 3), unit = 16 * blk + (lane & 15);
  return (k < hi && unit < ho) ? w[(size_t)k * ho + unit] : 0.f;
}
// hm == 0: two hidden layers (wm / bm unused)
__global__ void bbx_pmlp2_prepare_kernel(const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ wm,
                                         const float* __restrict__ bm, const float* __restrict__ w2, const float* __restrict__ b2,
                                         const float* __restrict__ wd, const float* __restrict__ bd, int cols, int h1, int hm, int h2,
                                         int HP1, int HPM, int HP2, float* __restrict__ out) {
  const int K1 = 4 * pmlp2_ks_for(cols), HPI = HPM ? HPM : HP1, hi = HPM ? hm : h1;
  const int o_b1 = K1 * HP1, o_am = o_b1 + HP1, o_a2 = o_am + HP1 * HPM, o_bm = o_a2 + HPI * HP2, o_b2 = o_bm + HPM, o_wd = o_b2 + HP2,
            o_bd = o_wd + HP2, total = o_bd + 4;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    float v = 0.f;
    if (i < o_b1) { const int k = i / HP1, h = i - k * HP1; v = (k < cols && h < h1) ? w1[(size_t)k * h1 + h] : 0.f; }
    else if (i < o_am) { const int h = i - o_b1; v = h < h1 ? b1[h] : 0.f; }
    else if (i < o_a2) v = pmlp2_perm(wm, h1, hm, HP1, i - o_am);
    else if (i < o_bm) v = pmlp2_perm(w2, hi, h2, HPI, i - o_a2);
    else if (i < o_b2) { const int h = i - o_bm; v = h < hm ? bm[h] : 0.f; }
    else if (i < o_wd) { const int h = i - o_b2; v = h < h2 ? b2[h] : 0.f; }
    else if (i < o_bd) { const int h = i - o_wd; v = h < h2 ? wd[h] : 0.f; }
    else v = i == o_bd ? bd[0] : 0.f;
    out[i] = v;
  }
}

// one hidden layer behind the first: hout = relu(b + A hin) (LAST = false) or the deciding layer's dot over it (LAST = true:
// returns this lane's share of the logit).  Two blocks of 16 units in flight, their A operands requested half a block pair ahead.
template <int NKI, int NKO, bool LAST>
__device__ __forceinline__ float pmlp2_hidden(const bbx_f32x4 (&hin)[NKI], bbx_f32x4* hout, const float* A, const float* bl, const float* wl,
                                              int lane, int lg4) {
  constexpr int HS = NKI / 2;
  float part = 0.f;
  auto pair = [&](int b2i) __attribute__((always_inline)) {
    bbx_f32x4 acc0 = *(const bbx_f32x4*)(bl + 16 * b2i + 4 * lg4), acc1 = *(const bbx_f32x4*)(bl + 16 * b2i + 16 + 4 * lg4);
    const bbx_f32x4* ap = (const bbx_f32x4*)A + (size_t)b2i * NKI * 64 + lane;
#pragma unroll
    for (int hf = 0; hf < 2; hf++) {
      bbx_f32x4 av0[HS], av1[HS];
#pragma unroll
      for (int q = 0; q < HS; q++) { av0[q] = ap[(hf * HS + q) * 64]; av1[q] = ap[(NKI + hf * HS + q) * 64]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < HS; q++) {
        const bbx_f32x4 hb = hin[hf * HS + q];
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
    acc0.x = acc0.x > 0.f ? acc0.x : 0.f; acc0.y = acc0.y > 0.f ? acc0.y : 0.f; acc0.z = acc0.z > 0.f ? acc0.z : 0.f; acc0.w = acc0.w > 0.f ? acc0.w : 0.f;
    acc1.x = acc1.x > 0.f ? acc1.x : 0.f; acc1.y = acc1.y > 0.f ? acc1.y : 0.f; acc1.z = acc1.z > 0.f ? acc1.z : 0.f; acc1.w = acc1.w > 0.f ? acc1.w : 0.f;
    if constexpr (LAST) {
      const bbx_f32x4 w0 = *(const bbx_f32x4*)(wl + 16 * b2i + 4 * lg4), w1v = *(const bbx_f32x4*)(wl + 16 * b2i + 16 + 4 * lg4);
      part = fmaf(acc0.x, w0.x, part); part = fmaf(acc0.y, w0.y, part); part = fmaf(acc0.z, w0.z, part); part = fmaf(acc0.w, w0.w, part);
      part = fmaf(acc1.x, w1v.x, part); part = fmaf(acc1.y, w1v.y, part); part = fmaf(acc1.z, w1v.z, part); part = fmaf(acc1.w, w1v.w, part);
    } else { hout[b2i] = acc0; hout[b2i + 1] = acc1; }
  };
  if constexpr (LAST) {
#pragma clang loop unroll(disable)
    for (int b2i = 0; b2i < NKO; b2i += 2) pair(b2i);
  } else {
#pragma unroll
    for (int b2i = 0; b2i < NKO; b2i += 2) pair(b2i);
  }
  return part;
}

template <int HP1, int HPM, int HP2, int KS, int NWAVES>
__global__ __launch_bounds__(NWAVES * WAVE, 16 / NWAVES) void bbx_pmlp2_act_kernel(const int32_t* __restrict__ obs, const int32_t* __restrict__ rows, int B,
                                                                             int obs_rows, int cols, const float* __restrict__ wp,
                                                                             const float* __restrict__ u, int32_t* __restrict__ actions,
                                                                             float* __restrict__ logprobs, int lgcap) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NK1 = HP1 / 16, NKM = HPM / 16, NK2 = HP2 / 16;    // blocks of 16 units per layer
  constexpr int HPI = HPM ? HPM : HP1, AMF = HP1 * HPM, A2F = AMF + HPI * HP2 + HPM;   // (A2F: everything in LDS in front of b2)
  float* a2 = (float*)smem;
  const float* W1p = wp;
  const float* b1p = wp + 4 * KS * HP1;
  const float* a2g = b1p + HP1;
  // (A2 | b2p | w3p are contiguous in the prepared buffer: one copy; the biases and the deciding weights of a block of units
  // are then LDS reads next to its A operands, not trips to memory in front of every block's MFMAs)
  // what the wave's own first tile needs from memory — its environment's row count and uniform number, the tile's rows — is
  // requested before the staging, not behind it (narrow rows only: the values wait in registers)
  constexpr bool PRE = KS == 3;
  float xa0[KS]; int n0 = 0; float uu0 = 0.f;
  if (PRE) {
    int e0 = (int)blockIdx.x * ((int)blockDim.x / WAVE) + (int)(threadIdx.x / WAVE);
    e0 = e0 < B ? e0 : B - 1;
    n0 = rows[e0]; uu0 = u[e0];
    int r = (int)(threadIdx.x & 15); r = r < obs_rows ? r : obs_rows - 1;
    const int32_t* xr = obs + ((size_t)e0 * obs_rows + r) * cols;
#pragma unroll
    for (int s = 0; s < KS; s++) {
      const int k = 4 * s + (int)((threadIdx.x & 63) >> 4);
      const int32_t xi = xr[k < cols ? k : 0];
      xa0[s] = k < cols ? (float)xi : 0.f;
    }
  }
  for (int i = (int)threadIdx.x; i < (A2F + 2 * HP2) / 4; i += (int)blockDim.x) ((bbx_f32x4*)a2)[i] = ((const bbx_f32x4*)a2g)[i];
  __syncthreads();
  const float* aml = a2;                                                      // middle layer (if any)
  const float* a2l = a2 + AMF;                                                // last hidden layer
  const float* bml = a2l + HPI * HP2;
  const float* b2l = a2 + A2F;
  const float* w3l = b2l + HP2;
  const int lane = lane_id(), wave = uni((int)(threadIdx.x / WAVE)), nw = (int)blockDim.x / WAVE;
  float* lg = a2 + A2F + 2 * HP2 + (size_t)wave * lgcap;                      // logits of this wave's environment
  const int lr = lane & 15, lg4 = lane >> 4;
  const float b3 = a2g[A2F + 2 * HP2];
  // A wave's tiles run one after the other, so with one environment per wave a launch lasts as long as its LARGEST pair set
  // (B = 4096: the mean is 1.7 tiles, the maximum 5-7; measured 60 us per launch for 25 us of matrix-core time, the same with
  // two or four waves per SIMD).  The wave therefore takes only the first tile of its environment itself; further tiles go
  // to a queue of the workgroup from which its waves help themselves (an LDS counter; logits land in the owner's LDS buffer),
  // and the owner samples after a barrier.
  float* lg_base = a2 + A2F + 2 * HP2;
  int* s_env = (int*)(lg_base + (size_t)nw * lgcap);                          // [nw] environment of each wave this round
  int* s_n = s_env + nw;                                                      // [nw] its row count
  int* s_q = s_n + nw;                                                        // queue length, and (s_q[1]) how much of it has been taken
  unsigned short* queue = (unsigned short*)(s_q + 2);                         // [nw * PMLP_MAXROWS / 16] (wave << 8) | tile
  for (int base = (int)blockIdx.x * nw; base < B; base += (int)gridDim.x * nw) {
    const int env = base + wave;
    const bool valid = env < B;
    const bool round0 = PRE && base == (int)blockIdx.x * nw;
    int n = valid ? uni(round0 ? n0 : rows[env]) : 0;
    const float uu = valid ? (round0 ? uu0 : u[env]) : 0.f;
    n = n < obs_rows ? n : obs_rows; n = n < PMLP_MAXROWS ? n : PMLP_MAXROWS; n = n > 0 ? n : 0;
    const int T = (n + 15) >> 4;
    __syncthreads();                                                          // (the previous round is over: logits, queue)
    if (lane == 0) { s_env[wave] = env; s_n[wave] = n; }
    if (threadIdx.x == 0) { s_q[0] = 0; s_q[1] = 0; }
    __syncthreads();
    for (int tl = lane; tl < T - 1; tl += WAVE) queue[atomicAdd(s_q, 1)] = (unsigned short)((wave << 8) | (1 + tl));   // (up to PMLP_MAXROWS / 16 = 128 tiles)
    __syncthreads();
    const int nq = uni(s_q[0]);
  {
    // the wave's own first tile, then tiles from the queue for as long as there are any: a wave whose environment has one tile
    // helps with the others' second and third at once (a barrier between "own" and "shared" tiles had the one-tile waves wait
    // 11 us per launch for the two-tile waves, with the matrix cores half idle)
    bool own = T > 0;
    for (;;) {
      int w = wave, t = 0;
      bool pre = false;
      if (own) { own = false; pre = round0; }
      else {
        int idx = 0;
        if (lane == 0) idx = atomicAdd(&s_q[1], 1);
        idx = __builtin_amdgcn_readfirstlane(idx);
        if (idx >= nq) break;
        const int e = uni((int)queue[idx]); w = e >> 8; t = e & 255;
      }
      const int tn = uni(s_n[w]), r0 = 16 * t;
      const int32_t* ob = obs + (size_t)uni(s_env[w]) * obs_rows * cols;
      float* lgt = lg_base + (size_t)w * lgcap;
    {
      int r = r0 + lr; r = r < obs_rows ? r : obs_rows - 1;                   // inside the block whatever the row count is
      const int32_t* xr = ob + (size_t)r * cols;
      float xa[KS];
      if (pre) {
#pragma unroll
        for (int s = 0; s < KS; s++) xa[s] = xa0[s];
      } else {
#pragma unroll
        for (int s = 0; s < KS; s++) {
          const int k = 4 * s + lg4;
          const int32_t xi = xr[k < cols ? k : 0];
          xa[s] = k < cols ? (float)xi : 0.f;
        }
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
      // ---- the hidden layers behind the first, then the deciding layer's dot
      float part;
      if constexpr (HPM != 0) {
        bbx_f32x4 hm[NKM];
        pmlp2_hidden<NK1, NKM, false>(h, hm, aml, bml, nullptr, lane, lg4);
        part = pmlp2_hidden<NKM, NK2, true>(hm, nullptr, a2l, b2l, w3l, lane, lg4);
      } else part = pmlp2_hidden<NK1, NK2, true>(h, nullptr, a2l, b2l, w3l, lane, lg4);
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

// padded layer sizes: two hidden layers are padded one by one; with a middle layer all three take the size of the widest
// (one kernel per size instead of eight)
static void pmlp2_pads(int h1, int hm, int h2, int* hp1, int* hpm, int* hp2) {
  if (hm == 0) { *hp1 = pmlp2_hp_for(h1); *hpm = 0; *hp2 = pmlp2_hp_for(h2); return; }
  const int mx = h1 > hm ? (h1 > h2 ? h1 : h2) : (hm > h2 ? hm : h2);
  *hp1 = *hpm = *hp2 = pmlp2_hp_for(mx);
}

extern "C" int bbx_pmlp2_floats(int cols, int h1, int hm, int h2) {
  int hp1, hpm, hp2; pmlp2_pads(h1, hm, h2, &hp1, &hpm, &hp2);
  return pmlp2_prepared_floats(cols, hp1, hpm, hp2);
}

extern "C" int bbx_launch_pmlp2_prepare(const float* w1, const float* b1, const float* wm, const float* bm, const float* w2, const float* b2,
                                        const float* wd, const float* bd, int cols, int h1, int hm, int h2, float* out, hipStream_t stream) {
  int hp1, hpm, hp2; pmlp2_pads(h1, hm, h2, &hp1, &hpm, &hp2);
  hipLaunchKernelGGL(bbx_pmlp2_prepare_kernel, dim3(64), dim3(256), 0, stream, w1, b1, wm, bm, w2, b2, wd, bd, cols, h1, hm, h2, hp1, hpm, hp2, out);
  return (int)hipGetLastError();
}

extern "C" int bbx_launch_pmlp2_act(const int32_t* obs, const int32_t* rows, int B, int obs_rows, int cols, const float* wp, int h1, int hm, int h2,
                                    const float* u, int32_t* actions, float* logprobs, int cus, int max_lds, hipStream_t stream) {
  int hp1, hpm, hp2; pmlp2_pads(h1, hm, h2, &hp1, &hpm, &hp2);
  const int ks = pmlp2_ks_for(cols);
  // 64 KB of second-layer weights: two workgroups of 8 waves per CU; with a middle layer of that size (128 KB): one workgroup
  // of 16 waves — or of 8 or 4 where tall observation blocks (4 bytes of logits per row and wave) leave less room
  int waves = (hpm == 128) ? 16 : PMLP2_WAVES;
  int lgcap = obs_rows < PMLP_MAXROWS ? obs_rows : PMLP_MAXROWS;              // logits per wave: what the block can hold
  lgcap = (lgcap + 63) / 64 * 64;
  size_t ml = 0;
  for (;; waves /= 2) {
    ml = ((size_t)hp1 * hpm + (size_t)(hpm ? hpm : hp1) * hp2 + hpm + 2 * hp2) * sizeof(float) + (size_t)waves * lgcap * sizeof(float) +
         (size_t)(2 * waves + 2) * sizeof(int) + (size_t)waves * (PMLP_MAXROWS / 16) * sizeof(unsigned short);
    if (ml <= (size_t)max_lds || hpm != 128 || waves == 4) break;
  }
  if (ml > (size_t)max_lds) return (int)hipErrorInvalidValue;
  int dev_ = 0; (void)hipGetDevice(&dev_); dev_ &= 63;
  const int max_blocks = (hpm == 128 ? 1 : 2) * (cus > 0 ? cus : 256);
  int blocks = (B + waves - 1) / waves;
  blocks = blocks < max_blocks ? blocks : max_blocks;
#define BBX_P2(N1, NM, N2, K, NW) do { \
    static std::atomic<size_t> set_[64];   /* per device, once per size: the call is not free (the attribute belongs to the current device's code object) */ \
    if (set_[dev_].load(std::memory_order_acquire) < ml) { \
      hipError_t err_ = hipFuncSetAttribute((const void*)bbx_pmlp2_act_kernel<N1, NM, N2, K, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ml); \
      if (err_ != hipSuccess) return (int)err_; \
      size_t old_ = set_[dev_].load(std::memory_order_relaxed); \
      while (old_ < ml && !set_[dev_].compare_exchange_weak(old_, ml, std::memory_order_release)) {} } \
    hipLaunchKernelGGL((bbx_pmlp2_act_kernel<N1, NM, N2, K, NW>), dim3(blocks), dim3(NW * WAVE), ml, stream, obs, rows, B, obs_rows, cols, wp, u, actions, logprobs, lgcap); } while (0)
#define BBX_P2_K(N1, NM, N2, NW) do { if (ks == 3) BBX_P2(N1, NM, N2, 3, NW); else if (ks == 8) BBX_P2(N1, NM, N2, 8, NW); else BBX_P2(N1, NM, N2, 16, NW); } while (0)
  if (hpm == 128 && waves == 16) BBX_P2_K(128, 128, 128, 16);
  else if (hpm == 128 && waves == 8) BBX_P2_K(128, 128, 128, 8);
  else if (hpm == 128) BBX_P2_K(128, 128, 128, 4);
  else if (hpm == 64) BBX_P2_K(64, 64, 64, 8);
  else if (hp1 == 64 && hp2 == 64) BBX_P2_K(64, 0, 64, 8); else if (hp1 == 64) BBX_P2_K(64, 0, 128, 8);
  else if (hp2 == 64) BBX_P2_K(128, 0, 64, 8); else BBX_P2_K(128, 0, 128, 8);
#undef BBX_P2_K
#undef BBX_P2
  return (int)hipGetLastError();
}
