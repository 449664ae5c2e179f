// ubench_xcd_sync.hip -- facts the XCD-tile kernel (dbcsr_amd/csrc/mm_tile.h) relies on, measured on the box:
//   [place]   where a persistent grid of 256 workgroups (one per CU: 512 threads, > 80 KB of LDS) lands: XCC_ID and HW_ID of
//             every workgroup -- is blockIdx % 8 the XCD, and does every workgroup get a CU of its own?
//   [window]  the progress-window protocol: every wave publishes a counter (agent-scope relaxed store) and reads the 256 counters
//             of its XCD team (one 1 KiB read per wave, agent-scope relaxed loads), minimum by DPP; cost per round, and how stale the
//             minimum a wave sees is (distance between its own counter and the minimum it reads when all waves run in step)
//   [latency] store -> visible to a wave on another CU of the same XCD / of another XCD (ping-pong, ns per hop)
// Development tool, not part of the product library:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ubench_xcd_sync ubench_xcd_sync.hip && ./ubench_xcd_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ void __launch_bounds__(512) place(unsigned* __restrict__ out) {
  extern __shared__ char smem[];
  unsigned x, h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = x;
    out[2 * blockIdx.x + 1] = h;
  }
  // stay resident for a while so that all workgroups of the grid coexist
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 20000) __builtin_amdgcn_s_sleep(10);
  if (threadIdx.x == 1000) smem[0] = 1;
}

__device__ __forceinline__ unsigned wave_min(unsigned v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned o = (unsigned)__shfl_xor((int)v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// MODE 0: agent-scope loads (sc1), 1: workgroup-scope loads (sc0), 2: plain loads
template <int MODE>
__global__ void __launch_bounds__(512) window(unsigned* __restrict__ prog /* 8 x 256 */, int rounds, int W, unsigned long long* __restrict__ stats) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int q = slot * 8 + wid;  // 0..255 in the team
  unsigned* team = prog + xcd * 256;
  unsigned long long waits = 0, lagsum = 0;
  const long long t0 = wall_clock64();
  unsigned seen = 0;
  for (int r = 1; r <= rounds; ++r) {
    // throttle: may not start round r before every team member finished round r - W
    int spins = 0;
    while ((int)r - (int)seen > W) {
      unsigned v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (MODE == 0) v[j] = __hip_atomic_load(team + lane * 4 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 1) v[j] = __hip_atomic_load(team + lane * 4 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 2) v[j] = reinterpret_cast<volatile unsigned*>(team)[lane * 4 + j];
      }
      unsigned m = v[0] < v[1] ? v[0] : v[1];
      const unsigned m2 = v[2] < v[3] ? v[2] : v[3];
      m = m < m2 ? m : m2;
      seen = wave_min(m);
      if ((int)r - (int)seen > W) {
        ++waits;
        if (++spins > (1 << 16)) { seen = 0x7fffffffu; if (lane == 0) atomicAdd(&stats[3], 1ull); }  // give up: never hang
        __builtin_amdgcn_s_sleep(2);
      }
    }
    lagsum += (unsigned)(r - (seen > (unsigned)r ? (unsigned)r : seen));
    // "work" of the round: a little, varying per wave
    const int work = 20 + ((q * 2654435761u + r * 40503u) >> 28);
    for (int i = 0; i < work; ++i) __builtin_amdgcn_s_sleep(1);
    if (lane == 0) __hip_atomic_store(team + q, (unsigned)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  const long long t1 = wall_clock64();
  if (lane == 0) {
    atomicAdd(&stats[0], (unsigned long long)(t1 - t0));
    atomicAdd(&stats[1], waits);
    atomicAdd(&stats[2], lagsum);
  }
  if (threadIdx.x == 1000) smem[0] = 1;
}

// ping-pong between workgroup 0 (wave 0) and workgroup `peer` (wave 0): ns per hop
__global__ void __launch_bounds__(512) pingpong(unsigned* __restrict__ flag, int peer, int hops, long long* __restrict__ out) {
  extern __shared__ char smem[];
  if (threadIdx.x >= 64) return;
  const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == peer ? 1 : -1);
  if (me < 0) return;
  const long long t0 = wall_clock64();
  int guard = 0;
  for (int h = 0; h < hops; ++h) {
    if ((h & 1) == me) {
      __hip_atomic_store(flag, (unsigned)(h + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(h + 1)) {
        if (++guard > (1 << 24)) break;
      }
    }
  }
  if (me == 0 && threadIdx.x == 0) out[0] = wall_clock64() - t0;
  if (threadIdx.x == 1000) smem[0] = 1;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int lds = 90 * 1024;  // one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(place), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(window<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(window<1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(window<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  int wcr = 0;
  CK(hipDeviceGetAttribute(&wcr, hipDeviceAttributeWallClockRate, 0));  // kHz
  printf("# wall clock %d kHz\n", wcr);
  {
    unsigned* d;
    CK(hipMalloc(&d, 256 * 2 * 4));
    for (int rep = 0; rep < 2; ++rep) {
      place<<<256, 512, lds>>>(d);
      CK(hipDeviceSynchronize());
      std::vector<unsigned> h(512);
      CK(hipMemcpy(h.data(), d, 512 * 4, hipMemcpyDeviceToHost));
      int ok = 0;
      std::vector<unsigned> ids;
      for (int b = 0; b < 256; ++b) {
        ok += (h[2 * b] & 0xf) == (unsigned)(b & 7);
        ids.push_back(((h[2 * b] & 0xf) << 16) | (h[2 * b + 1] & 0xff00u) | ((h[2 * b + 1] >> 13) & 7) << 4);  // xcc, cu_id (bits 8-11), sh (12), se (13-15)
      }
      std::sort(ids.begin(), ids.end());
      const int distinct = (int)(std::unique(ids.begin(), ids.end()) - ids.begin());
      printf("[place] launch %d: blockIdx %% 8 == XCC_ID for %d of 256 workgroups; %d distinct (xcc, se, sh, cu) places\n", rep, ok, distinct);
      if (rep == 0) {
        printf("[place] first 16 workgroups (blockIdx: xcc_id hw_id):");
        for (int b = 0; b < 16; ++b) printf(" %d:%u/%08x", b, h[2 * b] & 0xf, h[2 * b + 1]);
        printf("\n");
      }
    }
    CK(hipFree(d));
  }
  {
    unsigned* prog;
    unsigned long long* stats;
    CK(hipMalloc(&prog, 8 * 256 * 4));
    CK(hipMalloc(&stats, 4 * 8));
    const int rounds = 2000;
    for (int mode = 0; mode < 3; ++mode)
      for (int W : {1, 4, 16, 1000000}) {
        CK(hipMemset(prog, 0, 8 * 256 * 4));
        CK(hipMemset(stats, 0, 32));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (mode == 0) window<0><<<256, 512, lds>>>(prog, rounds, W, stats);
        if (mode == 1) window<1><<<256, 512, lds>>>(prog, rounds, W, stats);
        if (mode == 2) window<2><<<256, 512, lds>>>(prog, rounds, W, stats);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned long long h[4];
        CK(hipMemcpy(h, stats, 32, hipMemcpyDeviceToHost));
        printf("[window] loads %s  W %7d: %8.3f ms for %d rounds = %6.2f us per round; polls that had to wait %llu (%.2f per wave-round), mean lag seen %.2f rounds, gave up %llu\n",
               mode == 0 ? "agent(sc1)" : (mode == 1 ? "wg(sc0)   " : "plain     "), W, ms, rounds, 1e3 * ms / rounds, h[1], (double)h[1] / (2048.0 * rounds),
               (double)h[2] / (2048.0 * rounds), h[3]);
      }
    CK(hipFree(prog)); CK(hipFree(stats));
  }
  {
    unsigned* flag;
    long long* out;
    CK(hipMalloc(&flag, 4)); CK(hipMalloc(&out, 8));
    for (int peer : {8, 16, 1, 3, 255}) {
      CK(hipMemset(flag, 0, 4));
      const int hops = 2000;
      pingpong<<<256, 512, lds>>>(flag, peer, hops, out);
      CK(hipDeviceSynchronize());
      long long t;
      CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost));
      printf("[latency] workgroup 0 <-> %3d (%s XCD): %.0f ns per hop\n", peer, (peer & 7) == 0 ? "same" : "other", (double)t / wcr * 1e6 / hops);
    }
  }
  return 0;
}
