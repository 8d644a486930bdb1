// Instrumentation of the composite kernels for tools/ (count_visits.py, wave_timeline.py, bwd_timeline.py, deep phases): per-launch
// visit counters and per-wave start / end stamps.  Compiled to NOTHING unless the library is built with -DOMFS_DEBUG_COUNTERS or
// -DOMFS_DEBUG_TIMELINE (EXTRA_HIPCC_FLAGS of csrc/build.sh); the shipped library carries none of it.  Included inside namespace omfs.
#pragma once

#ifdef OMFS_DEBUG_COUNTERS
__device__ unsigned long long omfs_dbg[32];   // bwd: visits, visits with a hit, hit lanes; fwd: visits, visits with a hit, hit lanes
#define OMFS_DBG_ADD(i, v) do { if (lane_id() == 0) atomicAdd(&omfs_dbg[i], (unsigned long long)(v)); } while (0)
#else
#define OMFS_DBG_ADD(i, v) do { } while (0)
#endif
#ifdef OMFS_DEBUG_TIMELINE
// per-wave (workgroup for the deep forward) start / end on the 100 MHz real-time counter: tools/wave_timeline.py
constexpr int OMFS_DBG_TL = 1 << 19;
__device__ unsigned long long omfs_dbg_tl[3][2][OMFS_DBG_TL];
__device__ uint32_t omfs_dbg_work[3][OMFS_DBG_TL];       // splats visited by the wave
struct DbgSpan {
  int k; uint32_t i; unsigned long long t0; uint32_t work;
  __device__ DbgSpan(int k_, uint32_t i_) : k(k_), i(i_), t0(__builtin_amdgcn_s_memrealtime()), work(0) {}
  __device__ ~DbgSpan() {
    if (threadIdx.x == 0 && i < (uint32_t)OMFS_DBG_TL) {
      omfs_dbg_tl[k][0][i] = t0; omfs_dbg_tl[k][1][i] = __builtin_amdgcn_s_memrealtime(); omfs_dbg_work[k][i] = work;
    }
  }
};
#define OMFS_DBG_SPAN(k) DbgSpan omfs_dbg_span_(k, blockIdx.x)
#define OMFS_DBG_WORK() (++omfs_dbg_span_.work)
// composite_fwd only: shader-clock cycles per phase (0 waiting for the gather, 1 staging, 2 walking, 3 everything else)
__device__ uint32_t omfs_dbg_phase[4][OMFS_DBG_TL];
struct DbgPhase {
  unsigned long long t; uint32_t acc[4];
  __device__ DbgPhase() : t(__builtin_readcyclecounter()), acc{0u, 0u, 0u, 0u} {}
  __device__ void mark(int i) { const unsigned long long n = __builtin_readcyclecounter(); acc[i] += (uint32_t)(n - t); t = n; }
  __device__ ~DbgPhase() {
    mark(3);
    if (threadIdx.x == 0 && blockIdx.x < (uint32_t)OMFS_DBG_TL)
      for (int i = 0; i < 4; ++i) omfs_dbg_phase[i][blockIdx.x] = acc[i];
  }
};
#define OMFS_DBG_PHASES() DbgPhase omfs_dbg_phase_
#define OMFS_DBG_PHASE(i) omfs_dbg_phase_.mark(i)
// composite_fwd only: at the end of 64-entry step s (s < 8), 10 ns ticks since the wave started and entries walked so far
__device__ uint32_t omfs_dbg_step[2][8][OMFS_DBG_TL];
#define OMFS_DBG_STEP(s) do { if (threadIdx.x == 0 && blockIdx.x < (uint32_t)OMFS_DBG_TL && (s) < 8u) { \
    omfs_dbg_step[0][s][blockIdx.x] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - omfs_dbg_span_.t0); \
    omfs_dbg_step[1][s][blockIdx.x] = omfs_dbg_span_.work; } } while (0)
#define OMFS_DBG_WAIT_VM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define OMFS_DBG_SPAN(k) do { } while (0)
#define OMFS_DBG_WORK() do { } while (0)
#define OMFS_DBG_PHASES() do { } while (0)
#define OMFS_DBG_PHASE(i) do { } while (0)
#define OMFS_DBG_STEP(s) do { } while (0)
#define OMFS_DBG_WAIT_VM() do { } while (0)
#endif

// Host-side readers of the arrays above (each library that includes this header carries its own copy of the arrays and exports
// the readers under its own names: omfs_debug_* in libomfs_splat.so, omfs_experiment_debug_* in libomfs_experiments.so).
#ifdef OMFS_DEBUG_TIMELINE
static inline int dbg_timeline_read(int kernel, unsigned long long* out, int n, int reset) {   // out [2][n]
  OMFS_REQUIRE(kernel >= 0 && kernel < 3 && n > 0 && n <= OMFS_DBG_TL, "args");
  for (int e = 0; e < 2; ++e)
    OMFS_CHECK_HIP(hipMemcpyFromSymbol(out + (size_t)e * n, HIP_SYMBOL(omfs_dbg_tl), (size_t)n * 8,
                                       ((size_t)kernel * 2 + e) * OMFS_DBG_TL * 8));
  if (out && n > 0 && reset == 4) {      // reset == 4: out as uint32 [2][8][n] = per-step ticks and entries of composite_fwd
    for (int i = 0; i < 16; ++i)
      OMFS_CHECK_HIP(hipMemcpyFromSymbol((uint32_t*)out + (size_t)i * n, HIP_SYMBOL(omfs_dbg_step), (size_t)n * 4, (size_t)i * OMFS_DBG_TL * 4));
    void* p = nullptr;
    OMFS_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(omfs_dbg_step)));
    OMFS_CHECK_HIP(hipMemset(p, 0, sizeof(uint32_t) * 16 * OMFS_DBG_TL));
    return OMFS_OK;
  }
  if (out && n > 0 && reset == 3) {      // reset == 3: out [4][n/2] (as uint32 [4][n]) = the phase cycles of composite_fwd
    for (int i = 0; i < 4; ++i)
      OMFS_CHECK_HIP(hipMemcpyFromSymbol((uint32_t*)out + (size_t)i * n, HIP_SYMBOL(omfs_dbg_phase), (size_t)n * 4, (size_t)i * OMFS_DBG_TL * 4));
    return OMFS_OK;
  }
  if (out && n > 0 && reset >= 2) {      // reset == 2: out [n] also receives the work counters after the two time rows
    OMFS_CHECK_HIP(hipMemcpyFromSymbol(out + (size_t)2 * n, HIP_SYMBOL(omfs_dbg_work), (size_t)n * 4, (size_t)kernel * OMFS_DBG_TL * 4));
    return OMFS_OK;
  }
  if (reset) {
    void* p = nullptr;
    OMFS_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(omfs_dbg_tl)));
    OMFS_CHECK_HIP(hipMemset(p, 0, sizeof(unsigned long long) * 3 * 2 * OMFS_DBG_TL));
  }
  return OMFS_OK;
}

#endif
#ifdef OMFS_DEBUG_COUNTERS
static inline int dbg_counters_read(unsigned long long* out8, int reset) {
  OMFS_CHECK_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(omfs_dbg), 256));
  if (reset) { unsigned long long z[32] = {0}; OMFS_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(omfs_dbg), z, 256)); }
  return OMFS_OK;
}
#endif
