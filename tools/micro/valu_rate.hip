// Micro-benchmark: issue rate of wave64 VALU instructions on one SIMD as a function of resident waves.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_rate.hip -o gpurun_out/valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(x) x x x x x x x x
template <int KIND>
__global__ void rate_kernel(float* out, long long* cycles, int iters) {
  __shared__ float4 lds_pad[16];
  if (threadIdx.x < 16) lds_pad[threadIdx.x] = make_float4(1.f, 2.f, 3.f, 4.f);
  __syncthreads();
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  const float b = 1.0001f, c = 0.5f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (KIND == 0) {
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 1) {
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 2) {
      REP8(asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 3) {
      REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n"
                        "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");)
    } else if (KIND == 4) {
      REP8(asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                        "v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 5) {   // dependent chain: latency of one fma
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
                        "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
    } else if (KIND == 6) {   // v_readlane + s use
      REP8(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n"
                        "v_readlane_b32 s24, %4, 3\n v_readlane_b32 s25, %5, 5\n v_readlane_b32 s26, %6, 7\n v_readlane_b32 s27, %7, 9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s20","s21","s22","s23","s24","s25","s26","s27");)
    } else if (KIND == 7) {   // scalar only: eight independent 32-bit adds
      REP8(asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n"
                        "s_add_u32 s24, s24, 1\n s_add_u32 s25, s25, 1\n s_add_u32 s26, s26, 1\n s_add_u32 s27, s27, 1\n"
                        : : : "s20","s21","s22","s23","s24","s25","s26","s27","scc");)
    } else if (KIND == 8) {   // scalar only, the bit-scan idiom of the composite kernels: ffs, m-1 (two halves), and
      REP8(asm volatile("s_ff1_i32_b64 s28, s[20:21]\n s_add_u32 s22, s20, -1\n s_addc_u32 s23, s21, -1\n s_and_b64 s[20:21], s[20:21], s[22:23]\n"
                        "s_ff1_i32_b64 s29, s[24:25]\n s_add_u32 s26, s24, -1\n s_addc_u32 s27, s25, -1\n s_and_b64 s[24:25], s[24:25], s[26:27]\n"
                        : : : "s20","s21","s22","s23","s24","s25","s26","s27","s28","s29","scc");)
    } else if (KIND == 9) {   // eight vector fma + eight independent scalar adds, interleaved: sum or max of the two?
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n s_add_u32 s20, s20, 1\n v_fma_f32 %1, %1, %8, %9\n s_add_u32 s21, s21, 1\n"
                        "v_fma_f32 %2, %2, %8, %9\n s_add_u32 s22, s22, 1\n v_fma_f32 %3, %3, %8, %9\n s_add_u32 s23, s23, 1\n"
                        "v_fma_f32 %4, %4, %8, %9\n s_add_u32 s24, s24, 1\n v_fma_f32 %5, %5, %8, %9\n s_add_u32 s25, s25, 1\n"
                        "v_fma_f32 %6, %6, %8, %9\n s_add_u32 s26, s26, 1\n v_fma_f32 %7, %7, %8, %9\n s_add_u32 s27, s27, 1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                        : "s20","s21","s22","s23","s24","s25","s26","s27","scc");)
    } else if (KIND == 10) {  // compare -> scalar use of the mask (the ballot idiom): v_cmp to an SGPR pair, s_and, s_cmp
      REP8(asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n s_and_b64 s[22:23], s[20:21], exec\n v_cmp_lt_f32 s[24:25], %1, %8\n s_and_b64 s[26:27], s[24:25], exec\n"
                        "v_cmp_lt_f32 s[20:21], %2, %8\n s_and_b64 s[22:23], s[20:21], exec\n v_cmp_lt_f32 s[24:25], %3, %8\n s_and_b64 s[26:27], s[24:25], exec\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)
                        : "s20","s21","s22","s23","s24","s25","s26","s27","scc");)
    } else if (KIND == 11) {  // LDS: eight broadcast 16-byte reads (every lane the same address) per group
      REP8(asm volatile("ds_read_b128 v[40:43], %0\n ds_read_b128 v[44:47], %0 offset:16\n ds_read_b128 v[48:51], %0 offset:32\n ds_read_b128 v[52:55], %0 offset:48\n"
                        "ds_read_b128 v[40:43], %0 offset:64\n ds_read_b128 v[44:47], %0 offset:80\n ds_read_b128 v[48:51], %0 offset:96\n ds_read_b128 v[52:55], %0 offset:112\n"
                        "s_waitcnt lgkmcnt(0)\n"
                        : : "v"(0) : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","memory");)
    } else if (KIND == 13) {  // packed fp32 fma: two fmas per instruction -- at the price of one?
      REP8(asm volatile("v_pk_fma_f32 v[40:41], v[40:41], v[56:57], v[58:59]\n v_pk_fma_f32 v[42:43], v[42:43], v[56:57], v[58:59]\n"
                        "v_pk_fma_f32 v[44:45], v[44:45], v[56:57], v[58:59]\n v_pk_fma_f32 v[46:47], v[46:47], v[56:57], v[58:59]\n"
                        "v_pk_fma_f32 v[48:49], v[48:49], v[56:57], v[58:59]\n v_pk_fma_f32 v[50:51], v[50:51], v[56:57], v[58:59]\n"
                        "v_pk_fma_f32 v[52:53], v[52:53], v[56:57], v[58:59]\n v_pk_fma_f32 v[54:55], v[54:55], v[56:57], v[58:59]\n"
                        : : : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");)
    } else if (KIND == 14) {  // packed fp32 mul / add
      REP8(asm volatile("v_pk_mul_f32 v[40:41], v[40:41], v[56:57]\n v_pk_add_f32 v[42:43], v[42:43], v[56:57]\n"
                        "v_pk_mul_f32 v[44:45], v[44:45], v[56:57]\n v_pk_add_f32 v[46:47], v[46:47], v[56:57]\n"
                        "v_pk_mul_f32 v[48:49], v[48:49], v[56:57]\n v_pk_add_f32 v[50:51], v[50:51], v[56:57]\n"
                        "v_pk_mul_f32 v[52:53], v[52:53], v[56:57]\n v_pk_add_f32 v[54:55], v[54:55], v[56:57]\n"
                        : : : "v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55");)
    } else if (KIND == 15 || KIND == 16 || KIND == 17) {  // fma with a partial execution mask: does the SIMD skip rows (16 lanes) that are all off?
      const unsigned long long saved = __builtin_amdgcn_read_exec();
      const unsigned long long mk = KIND == 15 ? 0xFFFFull : (KIND == 16 ? 0xFFFFFFFFull : 0x0000FFFF0000FFFFull);
      asm volatile("s_mov_b64 exec, %0" : : "s"(mk));
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));)
      asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    } else if (KIND == 12) {  // s_cbranch that is never taken + scalar compare (loop-control idiom)
      REP8(asm volatile("s_cmp_eq_u32 s20, 77\n s_cbranch_scc1 1f\n s_cmp_eq_u32 s20, 78\n s_cbranch_scc1 1f\n"
                        "s_cmp_eq_u32 s20, 79\n s_cbranch_scc1 1f\n s_cmp_eq_u32 s20, 80\n s_cbranch_scc1 1f\n 1:\n"
                        : : : "s20","scc");)
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}

template <int KIND>
void run(const char* name, int waves_per_simd, float* out, long long* cyc) {
  const int iters = 2000;
  const int block = 64 * 4 * waves_per_simd > 1024 ? 1024 : 64 * 4 * waves_per_simd;   // one workgroup per CU carries 4*w waves
  const int blocks_per_cu = (64 * 4 * waves_per_simd) / block;
  const int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(block), 0, 0, out, cyc, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double n_inst = (double)iters * 64;   // per wave
  // wall: total wave-instructions per SIMD / time
  const double per_simd = n_inst * waves_per_simd;
  printf("%-10s waves/SIMD %d  wall %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cyc @2.4GHz); clock64 delta %lld ticks for %g inst\n",
         name, waves_per_simd, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, c, n_inst);
}

int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 16 * 1024 * 4); hipMalloc(&cyc, 8);
  for (int w : {1, 2, 4, 8}) {
    run<0>("fma", w, out, cyc);
    run<1>("exp", w, out, cyc);
    run<2>("add_dpp", w, out, cyc);
    run<3>("cmp+cnd", w, out, cyc);
    run<4>("mul/min", w, out, cyc);
    run<5>("fma_dep", w, out, cyc);
    run<6>("readlane", w, out, cyc);
    run<7>("s_add", w, out, cyc);
    run<8>("s_bitscan", w, out, cyc);
    run<9>("fma+s_add", w, out, cyc);
    run<10>("cmp->sgpr", w, out, cyc);
    run<11>("lds_b128", w, out, cyc);
    run<12>("cmp+branch", w, out, cyc);
    run<13>("pk_fma", w, out, cyc);
    run<15>("fma_16of64", w, out, cyc);
    run<16>("fma_32of64", w, out, cyc);
    run<17>("fma_rows0+2", w, out, cyc);
    run<14>("pk_mul/add", w, out, cyc);
  }
  return 0;
}
