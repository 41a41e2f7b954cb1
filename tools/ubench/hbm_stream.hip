// HBM streaming micro-benchmark (gfx950): what do pure streaming kernels reach on this box, and which access
// pattern gets there?  Sets the ceiling the HBM-bound kernels of the BC step (conv_bwd_prep, affine, pool, Adam)
// are priced against.   Build: hipcc --offload-arch=gfx950 -O3 -o hbm_stream hbm_stream.hip ; run: ./hbm_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// R input streams, W output streams (0/1), 16 B per lane per access.
// LAYOUT 0: grid-stride over the whole array (consecutive workgroups touch consecutive 4 KB);
// LAYOUT 1: each workgroup walks its own contiguous CHUNK bytes (like one (frame, channel-block) plane per workgroup)
template <int R, int W, int LAYOUT, int UNROLL, int NT>
__global__ __launch_bounds__(256) void stream_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, const u32x4* __restrict__ c,
                                                     u32x4* __restrict__ o, size_t n16, size_t chunk16, unsigned* sink) {
  u32x4 acc = {0, 0, 0, 0};
  size_t i, end, step;
  if (LAYOUT == 0) { i = (size_t)blockIdx.x * 256 + threadIdx.x; end = n16; step = (size_t)gridDim.x * 256; }
  else { i = (size_t)blockIdx.x * chunk16 + threadIdx.x; end = min(n16, (size_t)(blockIdx.x + 1) * chunk16); step = 256; }
#pragma unroll UNROLL
  for (; i < end; i += step) {
    u32x4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
    if (R > 1) v ^= NT ? __builtin_nontemporal_load(b + i) : b[i];
    if (R > 2) v ^= NT ? __builtin_nontemporal_load(c + i) : c[i];
    if (W) { if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v; }
    else acc ^= v;
  }
  if (!W && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) *sink = 1;
}
template <int LAYOUT, int NT>
__global__ __launch_bounds__(256) void fill_kernel(u32x4* __restrict__ o, size_t n16, size_t chunk16) {
  size_t i, end, step;
  if (LAYOUT == 0) { i = (size_t)blockIdx.x * 256 + threadIdx.x; end = n16; step = (size_t)gridDim.x * 256; }
  else { i = (size_t)blockIdx.x * chunk16 + threadIdx.x; end = min(n16, (size_t)(blockIdx.x + 1) * chunk16); step = 256; }
  const u32x4 v = {1, 2, 3, 4};
  for (; i < end; i += step) { if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v; }
}

static hipEvent_t e0, e1;
template <typename F> static double time_ms(F f, int reps = 5) {
  f(); hipDeviceSynchronize();
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0, 0); f(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t bytes = (size_t)2 << 30;   // 2 GiB per stream
  const size_t n16 = bytes / 16;
  u32x4 *a, *b, *c, *o; unsigned* sink;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&o, bytes); hipMalloc(&sink, 4);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 3, bytes); hipMemset(o, 0, bytes);
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("%-44s %9s %9s\n", "kernel", "ms", "TB/s");
#define RUN(NAME, STREAMS, ...) { double ms = time_ms([&] { __VA_ARGS__; }); printf("%-44s %9.3f %9.2f\n", NAME, ms, (STREAMS) * (double)bytes / ms / 1e9); }
  const size_t chunk = 256 * 1024 / 16;   // 256 KB per workgroup = one 64x64x32ch bf16 plane
  const unsigned gchunk = (unsigned)(n16 / chunk);
  for (unsigned wgs_per_cu : {4u, 8u, 16u, 32u}) {
    const unsigned g = 256 * wgs_per_cu;
    char nm[96];
    snprintf(nm, 96, "read1 gridstride u4 grid=%u", g);      RUN(nm, 1, (stream_kernel<1, 0, 0, 4, 0><<<g, 256>>>(a, b, c, o, n16, 0, sink)));
    snprintf(nm, 96, "copy  gridstride u4 grid=%u", g);      RUN(nm, 2, (stream_kernel<1, 1, 0, 4, 0><<<g, 256>>>(a, b, c, o, n16, 0, sink)));
    snprintf(nm, 96, "3r1w  gridstride u4 grid=%u", g);      RUN(nm, 4, (stream_kernel<3, 1, 0, 4, 0><<<g, 256>>>(a, b, c, o, n16, 0, sink)));
    snprintf(nm, 96, "fill  gridstride    grid=%u", g);      RUN(nm, 1, (fill_kernel<0, 0><<<g, 256>>>(o, n16, 0)));
  }
  RUN("3r1w gridstride u1 grid=4096", 4, (stream_kernel<3, 1, 0, 1, 0><<<4096, 256>>>(a, b, c, o, n16, 0, sink)));
  RUN("3r1w gridstride u2 grid=4096", 4, (stream_kernel<3, 1, 0, 2, 0><<<4096, 256>>>(a, b, c, o, n16, 0, sink)));
  RUN("3r1w gridstride u8 grid=4096", 4, (stream_kernel<3, 1, 0, 8, 0><<<4096, 256>>>(a, b, c, o, n16, 0, sink)));
  RUN("3r1w gridstride u4 NT grid=4096", 4, (stream_kernel<3, 1, 0, 4, 1><<<4096, 256>>>(a, b, c, o, n16, 0, sink)));
  RUN("copy gridstride u4 NT grid=4096", 2, (stream_kernel<1, 1, 0, 4, 1><<<4096, 256>>>(a, b, c, o, n16, 0, sink)));
  RUN("read1 gridstride u4 NT grid=4096", 1, (stream_kernel<1, 0, 0, 4, 1><<<4096, 256>>>(a, b, c, o, n16, 0, sink)));
  RUN("fill gridstride NT grid=4096", 1, (fill_kernel<0, 1><<<4096, 256>>>(o, n16, 0)));
  RUN("read1 plane/WG 256KB u2", 1, (stream_kernel<1, 0, 1, 2, 0><<<gchunk, 256>>>(a, b, c, o, n16, chunk, sink)));
  RUN("copy  plane/WG 256KB u2", 2, (stream_kernel<1, 1, 1, 2, 0><<<gchunk, 256>>>(a, b, c, o, n16, chunk, sink)));
  RUN("3r1w  plane/WG 256KB u2", 4, (stream_kernel<3, 1, 1, 2, 0><<<gchunk, 256>>>(a, b, c, o, n16, chunk, sink)));
  RUN("3r1w  plane/WG 256KB u4", 4, (stream_kernel<3, 1, 1, 4, 0><<<gchunk, 256>>>(a, b, c, o, n16, chunk, sink)));
  RUN("3r1w  plane/WG 256KB u4 NT", 4, (stream_kernel<3, 1, 1, 4, 1><<<gchunk, 256>>>(a, b, c, o, n16, chunk, sink)));
  RUN("3r1w  plane/WG 64KB u4", 4, (stream_kernel<3, 1, 1, 4, 0><<<gchunk * 4, 256>>>(a, b, c, o, n16, chunk / 4, sink)));
  RUN("3r1w  plane/WG 16KB u4", 4, (stream_kernel<3, 1, 1, 4, 0><<<gchunk * 16, 256>>>(a, b, c, o, n16, chunk / 16, sink)));
  RUN("2r1w  plane/WG 256KB u4", 3, (stream_kernel<2, 1, 1, 4, 0><<<gchunk, 256>>>(a, b, c, o, n16, chunk, sink)));
  RUN("hipMemcpyDtoD", 2, hipMemcpyAsync(o, a, bytes, hipMemcpyDeviceToDevice, 0));
  return 0;
}
