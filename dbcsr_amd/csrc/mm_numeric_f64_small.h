// mm_numeric_f64_small.h -- fp64, every block dimension of the multiply at most 8 (and not all C blocks within 4 x 4: mm_numeric_f64_tiny).
//
// The sizes libsmm_acc serves with its "tiny" dataflow (src/acc/libsmm_acc/kernels/smm_acc_dnt_tiny.h: the whole A and B block of a product in
// shared memory, one thread per C element; 5 x 5 x 5 ... 8 x 8 x 8 are tuned triplets of its parameter files).  Here: one wave per C block, the block
// ONE 8 x 8 tile of the 2 x 2 arrangement of v_mfma_f64_4x4x4_4b (two instructions per product: k in fours).  A block of at most 8 x 8 doubles is at
// most 512 bytes: ONE bounds-checked 8-byte buffer load per lane fetches all of A, a second all of B -- two VGPR pairs per product in flight (the
// exact-size kernels for 9 ... 32 carry 2 ... 16 registers per lane and operand).  D products are in flight per wave; measured, D = 2 is best: eight waves
// per SIMD hide the latency, a deeper ring only adds requests behind the end of a 14-product list.  Such a multiply is instruction issue, not arithmetic
// (128 ... 1024 flop per product): the records of up to 62 products come with one vector load (lane t holds record t), every lane prepares the byte
// addresses and counts of ITS record, and a request is six v_readlane into the two buffer resources -- the first form computed them on the scalar unit,
// 46 instructions per product on the ONE scalar unit a CU's waves share, and that was the whole run time.
// LDS: 1 KiB per wave, the blocks as they lie (A: column-major m x ks, B: ks x n); lanes past the end of a block receive zeros from the bounds check.
//
// Measured (tools/gpu_sessions/r06_37_small_blocks.sh; 1425 block rows, fill 0.1 -- the benchmark's structure at every size): profiles/r06_small_blocks.txt
// (5^3 ... 8^3: 5.25 -> 2.2-2.4 ms against the run-time-size kernel).
#ifndef DBCSR_AMD_MM_NUMERIC_F64_SMALL_H
#define DBCSR_AMD_MM_NUMERIC_F64_SMALL_H
#include "mm_numeric_f64.h"  // load_desc_uniform, LaneMap (smm_core.h), Desc / Entry (mm_types.h)
#include "mm_exact.h"        // IntC

namespace dbcsr_amd {

typedef unsigned int u32x2_small __attribute__((ext_vector_type(2)));

// one C block (position `pos` of the launch order) by one wave; la: the wave's 1 KiB of LDS
template <int D, bool WORK>
__device__ __forceinline__ void small_block(int64_t pos, int lane, double* la, const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                            const double* __restrict__ a_data, const double* __restrict__ b_data, double* __restrict__ c_out,
                                            const double* __restrict__ c_in, double alpha, double beta, int skip_empty, const int* __restrict__ order,
                                            const Work* __restrict__ work) {
  Desc d;
  Entry first = Entry::make(0, 0, 0);
  if constexpr (WORK) {
    // launch-order records (build_work): descriptor and first product in one read -- the chain order[] -> descs[] -> entries[] -> operands, four round
    // trips before the first MFMA of a wave that lives for fourteen products, becomes work[] -> operands
    const Work w = work[pos];
    d.prod_cnt = __builtin_amdgcn_readfirstlane(w.prod_cnt);
    if (d.prod_cnt < 0) return;  // padding position
    d.c_off = uniform64(w.c_off);
    d.cin_off = uniform64(w.cin_off);
    d.prod_start = uniform64(w.prod_start);
    const int mn = __builtin_amdgcn_readfirstlane(((int)(uint16_t)w.m) | (((int)(uint16_t)w.n) << 16));
    d.m = (int16_t)(mn & 0xffff);
    d.n = (int16_t)(mn >> 16);
    first.a_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.a_lo);
    first.b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)w.b_lo);
    first.w = d.prod_cnt > 0 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)w.w) : 0u;
  } else {
    const int cb = order[pos];
    if (cb < 0 || cb >= nblk) return;
    d = load_desc_uniform(descs, cb);
  }
  const int cnt = d.prod_cnt;
  if (skip_empty && cnt == 0) return;  // in-place accumulation (beta = 1): untouched blocks stay as they are
  const int m = d.m, n = d.n;
  const LaneMap L(lane);
  double* lb = la + 64;
  const int a_i0 = L.rowl + m * L.kq, a_i1 = a_i0 + 4 * m;  // <= 7 + 8 * 7: inside the image whatever m is (rows past m feed rows of C nobody stores)
  const int voff = lane * 8;
  const bool mine = L.rowd < m && L.coll < n;
  double cin = 0.0, acc = 0.0;
  // C_in is requested now, ahead of the whole product walk (and ahead of the records: the wait for them then never includes a request issued behind them)
  if (mine && d.cin_off >= 0) cin = c_in[d.cin_off + L.rowd + m * L.coll];
  u32x2_small ra[D], rb[D];
  // The scalar unit is what a CU's waves share: everything a product's two buffer resources are made of -- byte addresses, byte counts -- is computed by the
  // lanes for the records of a batch at once, and a request is six v_readlane into the resource registers.  (First form of this kernel: 46 scalar
  // instructions per product for offsets, shifts and 64-bit adds -- 2.5 ms of scalar issue per launch, the whole run time; profiles/r06_small_blocks.txt.)
  auto request = [&](int u, uint64_t pa, int abytes, uint64_t pb, int bbytes) {
    const __amdgpu_buffer_rsrc_t rsa = __builtin_amdgcn_make_buffer_rsrc((void*)pa, 0, abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc((void*)pb, 0, bbytes, 0x00020000);
    ra[u] = __builtin_amdgcn_raw_buffer_load_b64(rsa, voff, 0, 0);
    rb[u] = __builtin_amdgcn_raw_buffer_load_b64(rsb, voff, 0, 0);
    // the requests stay in program order: the wait in front of a product's LDS write counts the requests issued BEHIND its own (s_waitcnt vmcnt), and the
    // scheduler, left alone, issues the first round in reverse -- every later round then drains the pipeline at its second product
    __builtin_amdgcn_sched_barrier(0);
  };
  constexpr int NB = ((64 - D) / D) * D;  // records per batch: a multiple of D, and record s + D of a batch is still one of the 64 lanes (no bounds to test)
  // one batch of records; the first one (FIRST: a code path of its own, so that its waits are counted for it alone) starts from the launch-order record
  auto batch = [&](int base, auto first_tag) {
    constexpr bool FIRST = decltype(first_tag)::value != 0;
    const int nb = min(NB, cnt - base);
    const Entry own = entries[d.prod_start + base + (lane < nb ? lane : nb - 1)];  // (unconditional: nothing between this request and the next ones waits for it)
    if constexpr (FIRST && WORK) {
      // the first product comes with the launch-order record: asked for behind the records' load, so that waiting for them leaves these two in flight
      const int ks = first.ks();
      request(0, (uint64_t)(a_data + first.a_off()), m * ks * 8, (uint64_t)(b_data + first.b_off()), ks * n * 8);
    }
    // per lane: what the requests of ITS record are made of
    const int my_ks = lane < nb ? own.ks() : 0;  // k extent 0: a place past the end of the list asks for nothing and feeds zeros
    const uint64_t my_pa = (uint64_t)(a_data + own.a_off()), my_pb = (uint64_t)(b_data + own.b_off());
    const int pa_lo = (int)(uint32_t)my_pa, pa_hi = (int)(uint32_t)(my_pa >> 32), pb_lo = (int)(uint32_t)my_pb, pb_hi = (int)(uint32_t)(my_pb >> 32);
    const int my_ab = m * my_ks * 8, my_bb = my_ks * n * 8;
    auto issue = [&](int u, int s) {
      const uint64_t pa = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(pa_lo, s) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(pa_hi, s) << 32);
      const uint64_t pb = (uint64_t)(uint32_t)__builtin_amdgcn_readlane(pb_lo, s) | ((uint64_t)(uint32_t)__builtin_amdgcn_readlane(pb_hi, s) << 32);
      request(u, pa, __builtin_amdgcn_readlane(my_ab, s), pb, __builtin_amdgcn_readlane(my_bb, s));
    };
    if constexpr (!(FIRST && WORK)) issue(0, 0);
#pragma unroll
    for (int u = 1; u < D; ++u) issue(u, u);
    for (int g = 0; g < nb; g += D) {
#pragma unroll
      for (int u = 0; u < D; ++u) {
        const int s = g + u;  // (past the end of the list in the last round: k extent 0, zeros come and zeros are added -- no branch in the pipeline)
        const int ks = __builtin_amdgcn_readlane(my_ks, s);
        *reinterpret_cast<u32x2_small*>(la + lane) = ra[u];
        *reinterpret_cast<u32x2_small*>(lb + lane) = rb[u];
        issue(u, s + D);
        // (A needs no mask: a valid row's element at k >= ks lies past the m * ks doubles the load delivered -- zeros;
        //  B at k >= ks may be another column's element: 0 x finite = 0, but the data may hold anything -- masked)
        const int b_i = L.kq + __mul24(ks, L.coll);  // (a full-rate multiply: v_mul_lo_u32 takes four times as long)
        const double a0 = la[a_i0];
        double b0 = lb[b_i];
        b0 = L.kq < ks ? b0 : 0.0;
        acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b0, acc, 0, 0, 0);
        if (ks > 4) {
          const double a1 = la[a_i1];
          double b1 = lb[b_i + 4];
          b1 = L.kq + 4 < ks ? b1 : 0.0;
          acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b1, acc, 0, 0, 0);
        }
      }
    }
  };
  if (cnt > 0) batch(0, IntC<1>{});
  for (int base = NB; base < cnt; base += NB) batch(base, IntC<0>{});
  if (mine) c_out[d.c_off + L.rowd + m * L.coll] = alpha * acc + (d.cin_off >= 0 ? beta * cin : 0.0);
}

// A wave takes G consecutive positions of the launch order, one after the other.  With SHORT product lists (one or two products per C block: a sparse product)
// a wave lives for about a microsecond and the launch is bound by the rate at which waves start: 5 x 5 blocks at 1 % fill (14 M C blocks) 4.96 ms with G = 1,
// 4.2 with G = 8 ... 16; with fourteen products per C block G makes no difference (2.21 ms either way): the host passes 8 or 1 (session r06_63).
// (Counters of the final form at fourteen products per block, session r06_62: LDS 0.46, scalar unit 0.31, VALU 0.18 busy -- no pipe is the bound, the dependent
//  chain of a product is: wait for the operands, LDS write, LDS read, two dependent MFMAs.)
template <int D, bool WORK>
__global__ void __launch_bounds__(256) mm_numeric_f64_small(const Desc* __restrict__ descs, int64_t nblk, const Entry* __restrict__ entries,
                                                            const double* __restrict__ a_data, const double* __restrict__ b_data,
                                                            double* __restrict__ c_out, const double* __restrict__ c_in, double alpha, double beta,
                                                            int skip_empty, const int* __restrict__ order, const Work* __restrict__ work, int G,
                                                            int64_t npos) {
  __shared__ double smem[4][128];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t pos0 = ((int64_t)wg * 4 + wid) * G;
  for (int g = 0; g < G; ++g) {
    const int64_t pos = pos0 + g;
    if (pos >= npos) break;
    small_block<D, WORK>(pos, lane, smem[wid], descs, nblk, entries, a_data, b_data, c_out, c_in, alpha, beta, skip_empty, order, work);
  }
}

}  // namespace dbcsr_amd
#endif
