// Micro-benchmark (gfx950): cost of v_permlane32_swap / v_permlane16_swap next to plain VALU, (a) as a dependent chain, (b) as independent
// streams, one and two waves per SIMD.  Question (round 4): the residual path of vpt_conv3x3_kernel's epilogue is 128 swaps + 128 plain ops per
// wave and costs 16 % of a K = 1152 tile -- are the swaps expensive?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/permlane.hip -o tools/ubench/permlane && tools/ubench/permlane
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32;

template <int KIND, int CHAINS>
__global__ __launch_bounds__(256) void k(u32* out, int iters) {
  u32 a[CHAINS], b[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { a[c] = threadIdx.x * 3 + c; b[c] = threadIdx.x * 7 + c; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      if (KIND == 0) { a[c] = a[c] * 3 + b[c]; b[c] = b[c] ^ a[c]; }                                     // 2 plain VALU (mad, xor)
      else if (KIND == 1) { auto s = __builtin_amdgcn_permlane32_swap(a[c], b[c], false, false); a[c] = s[0]; b[c] = s[1]; }
      else if (KIND == 2) { auto s = __builtin_amdgcn_permlane16_swap(a[c], b[c], false, false); a[c] = s[0]; b[c] = s[1]; }
      else if (KIND == 3) { a[c] = __shfl_xor(a[c], 32, 64); b[c] ^= a[c]; }                              // ds_bpermute / dpp path the compiler picks
      else if (KIND == 4) { a[c] = (u32)__builtin_amdgcn_update_dpp(0, (int)a[c], 0x124 /* row_ror:4 */, 0xf, 0xf, false); b[c] ^= a[c]; }
    }
  }
  u32 s = 0;
  for (int c = 0; c < CHAINS; ++c) s += a[c] ^ b[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND, int CHAINS>
static void run(u32* out, const char* name, int wg_per_cu) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND, CHAINS><<<256 * wg_per_cu, 256>>>(out, 100);
  hipEventRecord(e0);
  k<KIND, CHAINS><<<256 * wg_per_cu, 256>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per = ms * 1e6 / ((double)iters * CHAINS);   // ns per chain-iteration per wave (waves of one SIMD run concurrently)
  printf("%-34s chains=%d  %d wave(s)/SIMD: %7.2f ns per op-group per wave  (%.1f cycles at 2.1 GHz)\n", name, CHAINS, wg_per_cu, per, per * 2.1);
}

int main() {
  u32* out; hipMalloc(&out, 1024 * 1024 * sizeof(u32));
  for (int w = 1; w <= 2; ++w) {
    run<0, 1>(out, "2 plain VALU (dependent)", w);       run<0, 8>(out, "2 plain VALU", w);
    run<1, 1>(out, "v_permlane32_swap (dependent)", w);  run<1, 8>(out, "v_permlane32_swap", w);
    run<2, 1>(out, "v_permlane16_swap (dependent)", w);  run<2, 8>(out, "v_permlane16_swap", w);
    run<3, 1>(out, "__shfl_xor 32 + xor (dependent)", w); run<3, 8>(out, "__shfl_xor 32 + xor", w);
    run<4, 1>(out, "dpp row_ror:4 + xor (dependent)", w); run<4, 8>(out, "dpp row_ror:4 + xor", w);
  }
  return 0;
}
