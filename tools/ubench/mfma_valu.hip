// Micro-benchmark (gfx950): how much VALU work fits into the shadow of back-to-back MFMA 32x32x16 bf16, (a) inside one wave,
// (b) from a second wave on the same SIMD.  Decides whether an epilogue can be hidden behind another tile's MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu && tools/ubench/mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// K independent VALU ops (plain v_fma_f32, or v_pk_fma_f32 when PK) after every MFMA of the same wave
template <int K, bool PK>
__global__ __launch_bounds__(256) void same_wave(float* out, int iters) {
  extern __shared__ char lds[];
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)1.0f; }
  f32x2 v[16];
  for (int k = 0; k < 16; ++k) v[k] = f32x2{(float)k, (float)threadIdx.x};
  const f32x2 m = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if (PK) v[k] = v[k] * m + c; else v[k].x = fmaf(v[k].x, m.x, c.x);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  for (int k = 0; k < 16; ++k) s += v[k].x + v[k].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 8 waves per workgroup = 2 per SIMD: waves 0-3 issue MFMAs only (n_mfma of them), waves 4-7 VALU only (n_valu of them);
// each role reports its own duration (100 MHz wall clock)
template <int PRIO_VALU, int PRIO_MFMA>
__global__ __launch_bounds__(512) void two_waves(float* out, long long* t, int n_mfma, int n_valu) {
  extern __shared__ char lds[];
  const int w = threadIdx.x >> 6;
  float s = 0.f;
  const long long t0 = wall_clock64();
  if (w < 4) {
    __builtin_amdgcn_s_setprio(PRIO_MFMA);
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 7); b[e] = (__bf16)1.0f; }
    for (int i = 0; i < n_mfma / 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  } else {
    __builtin_amdgcn_s_setprio(PRIO_VALU);
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = (float)(k + threadIdx.x);
    for (int i = 0; i < n_valu / 8; ++i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = fmaf(v[k], 1.0001f, 0.5f);
    }
    for (int k = 0; k < 8; ++k) s += v[k];
  }
  const long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) t[blockIdx.x * 8 + w] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, bool PK>
static void run_same(float* out, int wgs_per_cu) {
  const int iters = 20000, lds = wgs_per_cu == 1 ? 100 * 1024 : 60 * 1024;
  hipFuncSetAttribute((const void*)same_wave<K, PK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  same_wave<K, PK><<<256 * wgs_per_cu, 256, lds>>>(out, 100);
  hipEventRecord(e0);
  same_wave<K, PK><<<256 * wgs_per_cu, 256, lds>>>(out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mf = (double)iters * 4 * wgs_per_cu;   // MFMAs per SIMD
  printf("same wave, %d wave(s)/SIMD, %2d %s per MFMA: %7.1f ns per MFMA per SIMD  (%.0f TF/s)\n", wgs_per_cu, K, PK ? "v_pk_fma_f32" : "v_fma_f32    ",
         ms * 1e6 / mf, 2.0 * 32 * 32 * 16 * mf * 1024 / (ms * 1e-3) / 1e12);
}

template <int PV, int PM>
static void run_two(float* out, long long* t) {
  const int lds = 100 * 1024;
  hipFuncSetAttribute((const void*)two_waves<PV, PM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  std::vector<long long> h(256 * 8);
  const int cases[][2] = {{40000, 0}, {0, 320000}, {40000, 320000}, {40000, 160000}, {40000, 80000}};
  for (auto& c : cases) {
    two_waves<PV, PM><<<256, 512, lds>>>(out, t, c[0], c[1]);
    two_waves<PV, PM><<<256, 512, lds>>>(out, t, c[0], c[1]);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), t, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double tm = 0, tv = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? tm : tv) += h[b * 8 + w] * 10.0;  // ns
    tm /= 1024; tv /= 1024;
    printf("two waves/SIMD (prio valu %d mfma %d): %6d MFMA | %6d VALU:  MFMA wave %8.1f us (%5.1f ns/MFMA)   VALU wave %8.1f us (%5.2f ns/op)\n", PV, PM, c[0], c[1],
           tm / 1e3, c[0] ? tm / c[0] : 0.0, tv / 1e3, c[1] ? tv / c[1] : 0.0);
  }
}

int main() {
  float* out; long long* t;
  hipMalloc(&out, 512 * 512 * sizeof(float)); hipMalloc(&t, 256 * 8 * sizeof(long long));
  run_same<0, false>(out, 1); run_same<2, false>(out, 1); run_same<4, false>(out, 1); run_same<6, false>(out, 1); run_same<8, false>(out, 1);
  run_same<12, false>(out, 1); run_same<16, false>(out, 1);
  run_same<4, true>(out, 1); run_same<8, true>(out, 1);
  run_same<0, false>(out, 2); run_same<4, false>(out, 2); run_same<8, false>(out, 2); run_same<16, false>(out, 2);
  run_two<0, 0>(out, t);
  run_two<3, 0>(out, t);
  run_two<0, 3>(out, t);
  return 0;
}
