// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes of this path, on KNOWN byte counts.
// MI355X_MICROARCH.md establishes "FETCH_SIZE reports half of the bytes" only for wide coalesced streaming reads and says
// other access widths are uncalibrated; the composite kernels GATHER 16-byte records by sorted id.  Kernels (each reads a
// table far larger than the 256 MiB Infinity Cache exactly once, so every byte comes from HBM):
//   stream16   : 16 B per lane, fully coalesced (the Adam pass / the calibrated case)
//   gather16   : 16 B per lane at uniformly random 16-byte-aligned records (g0 / g1 / g2 gathers of composite_*)
//   gather16x3 : three 16-byte records of the same random index from three tables (one staged list entry)
//   gather64   : 64-byte records, 16 lanes per record (dface / dsplat rows, checkpoints)
//   atomic4    : float atomic adds, 16 lanes per 64-byte record at random rows (the backward's flush)
// build : hipcc --offload-arch=gfx950 -O3 tools/micro/fetch_calib.hip -o /tmp/fetch_calib
// run   : rocprofv3 --pmc FETCH_SIZE --kernel-trace -d <dir> -- /tmp/fetch_calib     (and again with WRITE_SIZE)
//         python tools/micro/fetch_calib_report.py <dir_fetch> <dir_write>  ->  bytes the kernel really moved vs counter * 1024
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void stream16(const float4* __restrict__ t, size_t n, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = t[i];
  if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = 1.f;
}
__global__ void stream4(const float* __restrict__ t, size_t n, float* out) {        // 4 B per lane, coalesced: the planar [59][n_pad] reads
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (t[i] == 12345.678f) out[0] = 1.f;
}
__global__ void stream8(const float2* __restrict__ t, size_t n, float* out) {       // 8 B per lane, coalesced: (depth, id) keys
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float2 v = t[i];
  if (v.x + v.y == 12345.678f) out[0] = 1.f;
}
__global__ void gather16(const float4* __restrict__ t, const uint32_t* __restrict__ idx, size_t n, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float4 v = t[idx[i]];
  if (v.x + v.y + v.z + v.w == 12345.678f) out[0] = 1.f;
}
__global__ void gather16x3(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c,
                           const uint32_t* __restrict__ idx, size_t n, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t j = idx[i];
  const float4 v = a[j], w = b[j];
  const float x = c[j].x;
  if (v.x + v.y + w.z + w.w + x == 12345.678f) out[0] = 1.f;
}
__global__ void gather64(const float* __restrict__ t, const uint32_t* __restrict__ idx, size_t n_rec, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // 16 lanes per record
  if ((i >> 4) >= n_rec) return;
  const float v = t[(size_t)(idx[i >> 4] >> 2) * 16 + (i & 15)];       // 64-byte records: a quarter as many as 16-byte ones
  if (v == 12345.678f) out[0] = 1.f;
}
__global__ void atomic4(float* __restrict__ t, const uint32_t* __restrict__ idx, size_t n_rec) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;     // 16 lanes per record, 9 of them add (one dsplat flush)
  if ((i >> 4) >= n_rec || (i & 15) >= 9) return;
  atomicAdd(&t[(size_t)(idx[i >> 4] >> 2) * 16 + (i & 15)], 1.0f);
}

int main() {
  const size_t table_bytes = (size_t)2 << 30;               // 2 GiB per table: 8 x the Infinity Cache
  const size_t n16 = table_bytes / 16, n_idx = n16 / 4;     // every kernel touches 512 MiB of its table
  float4 *ta, *tb, *tc;
  uint32_t* idx;
  float* out;
  CHECK(hipMalloc(&ta, table_bytes)); CHECK(hipMalloc(&tb, table_bytes)); CHECK(hipMalloc(&tc, table_bytes));
  CHECK(hipMalloc(&idx, n_idx * 4)); CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(ta, 0, table_bytes)); CHECK(hipMemset(tb, 0, table_bytes)); CHECK(hipMemset(tc, 0, table_bytes));
  uint32_t* h = (uint32_t*)malloc(n_idx * 4);
  uint64_t s = 88172645463325252ull;
  for (size_t i = 0; i < n_idx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % n16); }
  CHECK(hipMemcpy(idx, h, n_idx * 4, hipMemcpyHostToDevice));
  const int B = 256;
  for (int rep = 0; rep < 3; ++rep) {
    stream16<<<(unsigned)((n_idx + B - 1) / B), B>>>(ta + (size_t)rep * n_idx, n_idx, out);
    stream4<<<(unsigned)((n_idx * 4 + B - 1) / B), B>>>((const float*)(tb + (size_t)rep * n_idx), n_idx * 4, out);
    stream8<<<(unsigned)((n_idx * 2 + B - 1) / B), B>>>((const float2*)(tc + (size_t)rep * n_idx), n_idx * 2, out);
    gather16<<<(unsigned)((n_idx + B - 1) / B), B>>>(ta, idx, n_idx, out);
    gather16x3<<<(unsigned)((n_idx + B - 1) / B), B>>>(ta, tb, tc, idx, n_idx, out);
    gather64<<<(unsigned)((n_idx / 4 * 16 + B - 1) / B), B>>>((const float*)tb, idx, n_idx / 4, out);
    atomic4<<<(unsigned)((n_idx / 4 * 16 + B - 1) / B), B>>>((float*)tc, idx, n_idx / 4);
    CHECK(hipDeviceSynchronize());
  }
  // ground truth per launch (bytes): what the kernel's loads / stores ask for, index reads included
  printf("KNOWN stream16 read %zu write 0\n", n_idx * 16);
  printf("KNOWN stream4 read %zu write 0\n", n_idx * 16);
  printf("KNOWN stream8 read %zu write 0\n", n_idx * 16);
  printf("KNOWN gather16 read %zu write 0\n", n_idx * 16 + n_idx * 4);
  printf("KNOWN gather16x3 read %zu write 0\n", n_idx * 48 + n_idx * 4);
  printf("KNOWN gather64 read %zu write 0\n", (n_idx / 4) * 64 + (n_idx / 4) * 4);
  printf("KNOWN atomic4 read %zu write %zu\n", (n_idx / 4) * 4, (n_idx / 4) * 36);
  return 0;
}
