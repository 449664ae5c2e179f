// mm_workspace.h -- device work areas (DevBuf) and the exclusive scan of the engine
// Part of the device-resident multiply engine: included by mm_engine.hip (one translation unit), in this order:
// mm_workspace.h, mm_symbolic.h, mm_numeric_f64.h, mm_numeric_f32.h, mm_aux.h.
#ifndef DBCSR_AMD_MM_WORKSPACE_H
#define DBCSR_AMD_MM_WORKSPACE_H

namespace dbcsr_amd {

// ----------------------------------------------------------------------------
// small utilities
// ----------------------------------------------------------------------------
// lab build, DBCSR_AMD_MM_POISON=<byte>: every work area is filled with that byte when it is allocated, so that a kernel reading a
// word nobody wrote computes with garbage EVERY time instead of with whatever the allocation held before (tests/test_gpu_poison.py)
static int g_devbuf_poison = -1;

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  int ensure(size_t n) {
    if (n <= cap) return 0;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 64;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
    if (e != hipSuccess) return check(e, "hipMalloc(workspace)", __FILE__, __LINE__);
    cap = want;
    if (g_devbuf_poison >= 0) (void)hipMemset(p, g_devbuf_poison, want * sizeof(T));
    return 0;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};


// ----------------------------------------------------------------------------
// exclusive scan (int32 in -> TO out), three small kernels
// ----------------------------------------------------------------------------
constexpr int kScanThreads = 256;
constexpr int kScanItems = 16;
constexpr int kScanChunk = kScanThreads * kScanItems;

__device__ __forceinline__ int64_t block_exclusive_scan(int64_t v, int64_t* total) {
  __shared__ int64_t wsum[kScanThreads / 64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int64_t inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) wsum[w] = inc;
  __syncthreads();
  int64_t woff = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < kScanThreads / 64; ++i) {
    if (i < w) woff += wsum[i];
    tot += wsum[i];
  }
  __syncthreads();
  *total = tot;
  return woff + inc - v;
}

__global__ void __launch_bounds__(kScanThreads) scan_reduce(const int* __restrict__ in, int64_t n, int64_t* __restrict__ partial) {
  const int64_t base = (int64_t)blockIdx.x * kScanChunk;
  int64_t s = 0;
  for (int it = 0; it < kScanItems; ++it) {
    const int64_t i = base + (int64_t)it * kScanThreads + threadIdx.x;
    if (i < n) s += in[i];
  }
  int64_t tot;
  (void)block_exclusive_scan(s, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(kScanThreads) scan_partials(int64_t* __restrict__ partial, int np, int64_t* __restrict__ total_out) {
  int64_t carry = 0;
  for (int base = 0; base < np; base += kScanThreads) {
    const int i = base + threadIdx.x;
    const int64_t v = i < np ? partial[i] : 0;
    int64_t tot;
    const int64_t ex = block_exclusive_scan(v, &tot);
    if (i < np) partial[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0 && total_out) *total_out = carry;
}

template <typename TO>
__global__ void __launch_bounds__(kScanThreads) scan_apply(const int* __restrict__ in, int64_t n, const int64_t* __restrict__ partial,
                                                            TO* __restrict__ out, int write_total_at_n) {
  const int64_t base = (int64_t)blockIdx.x * kScanChunk;
  // thread t owns kScanItems consecutive items
  const int64_t first = base + (int64_t)threadIdx.x * kScanItems;
  int v[kScanItems];
  int64_t s = 0;
#pragma unroll
  for (int it = 0; it < kScanItems; ++it) {
    const int64_t i = first + it;
    v[it] = i < n ? in[i] : 0;
    s += v[it];
  }
  int64_t tot;
  int64_t ex = block_exclusive_scan(s, &tot) + partial[blockIdx.x];
#pragma unroll
  for (int it = 0; it < kScanItems; ++it) {
    const int64_t i = first + it;
    if (i < n) out[i] = (TO)ex;
    ex += v[it];
    if (write_total_at_n && i == n - 1) out[n] = (TO)ex;
  }
}

}  // namespace dbcsr_amd
#endif
