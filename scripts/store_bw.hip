// Store-bandwidth ceiling for the observation pattern of the HBM-resident classes: W waves (one per environment, 4 per
// SIMD at W = 4096), each writing `bytes` contiguous bytes at env * stride with 16-byte-per-lane stores, `iters` times
// (the observation block is rewritten every step).  Variant 1 adds what the step kernel cannot avoid: one dependent
// load behind every block of stores (gfx9 retires loads and stores through the same in-order counter).
//   hipcc --offload-arch=gfx950 -O3 scripts/store_bw.hip -o scripts/_build/store_bw && scripts/_build/store_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int DEP>
__global__ __launch_bounds__(256, 4) void store_kernel(uint4* out, const unsigned* chase, size_t stride16, int n16, int iters, unsigned* sink) {
  int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  uint4* p = out + (size_t)wave * stride16;
  unsigned acc = wave;
  for (int it = 0; it < iters; it++) {
    uint4 v = make_uint4(acc, it, lane, wave);
    for (int i = lane; i < n16; i += 64) p[i] = v;
    if (DEP) acc = chase[(acc + it) & 0xffff];          // a dependent load behind the stores
  }
  if (acc == 0xdeadbeef) *sink = acc;
}

int main(int argc, char** argv) {
  int W = argc > 1 ? atoi(argv[1]) : 4096, iters = argc > 2 ? atoi(argv[2]) : 200;
  size_t stride = 2048 * 80;                            // obs_rows cap 2048 x 20 int32
  uint4* out; unsigned *chase, *sink;
  CK(hipMalloc(&out, (size_t)W * stride)); CK(hipMalloc(&chase, 65536 * 4)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(chase, 0, 65536 * 4));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int bytes : {4096, 20480, 81920}) {
    for (int dep = 0; dep < 2; dep++) {
      for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a));
        if (dep) hipLaunchKernelGGL(store_kernel<1>, dim3(W / 4), dim3(256), 0, 0, out, chase, stride / 16, bytes / 16, iters, sink);
        else hipLaunchKernelGGL(store_kernel<0>, dim3(W / 4), dim3(256), 0, 0, out, chase, stride / 16, bytes / 16, iters, sink);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (rep) printf("waves %d  bytes/wave/iter %6d  dependent_load %d : %.3f ms  %.2f TB/s  %.2f us per iteration\n", W, bytes, dep, ms,
                        (double)W * bytes * iters / (ms * 1e-3) / 1e12, ms * 1e3 / iters);
      }
    }
  }
  return 0;
}
