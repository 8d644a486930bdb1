// Micro-benchmark: what does a launch of workgroups that exit at once cost?  One-wave workgroups (with and without a few KB of
// static LDS), 512-thread workgroups with 34 KB of LDS (the shape of the segment-parallel forward), with one scalar load + compare
// in front of the exit.  Kernel time by HIP events over 20 launches.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/dispatch_rate.hip -o gpurun_out/dispatch_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int LDS_WORDS>
__global__ void exit_kernel(const unsigned* __restrict__ limit, unsigned* __restrict__ out) {
  __shared__ unsigned pad[LDS_WORDS > 0 ? LDS_WORDS : 1];
  if (blockIdx.x >= limit[0]) return;
  if (LDS_WORDS > 0) pad[threadIdx.x % LDS_WORDS] = threadIdx.x;
  __syncthreads();
  out[blockIdx.x] = LDS_WORDS > 0 ? pad[(threadIdx.x + 1) % LDS_WORDS] : threadIdx.x;
}

template <int LDS_WORDS>
static float time_launch(int grid, int block, const unsigned* limit, unsigned* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(exit_kernel<LDS_WORDS>, dim3(grid), dim3(block), 0, 0, limit, out);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(exit_kernel<LDS_WORDS>, dim3(grid), dim3(block), 0, 0, limit, out);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  return ms / 20.f * 1e3f;
}

int main() {
  unsigned *limit, *out;
  hipMalloc(&limit, 4); hipMalloc(&out, 4 << 20);
  hipMemset(limit, 0, 4);
  const int grids[] = {1, 8192, 32768, 131072, 294912, 589824};
  printf("%10s %14s %14s %18s %18s\n", "workgroups", "64 thr, no LDS", "64 thr, 3 KB", "512 thr, 34 KB", "1024 thr, 34 KB");
  for (int g : grids) {
    const float t0 = time_launch<0>(g, 64, limit, out), t1 = time_launch<768>(g, 64, limit, out);
    const float t2 = time_launch<8704>(g, 512, limit, out), t3 = time_launch<8704>(g, 1024, limit, out);
    printf("%10d %11.1f us %11.1f us %15.1f us %15.1f us\n", g, t0, t1, t2, t3);
  }
  return 0;
}
