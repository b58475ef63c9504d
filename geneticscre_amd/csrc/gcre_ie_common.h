// gcre_ie_common.h -- device helpers shared by the inclusion-exclusion null kernels (gcre_ie.hip, gcre_ieq.hip):
// wave reductions, the per-XCD work queues of the pruned kernels, the one-instruction bit adders.
#pragma once
#include "gcre_bitslice.h"
#include "gcre_kernels.h"

namespace gcre {

constexpr int kIeWaves = 4;
constexpr int kIeQueueStride = 16;  // words between two ticket counters (64 B)
constexpr int kIeRefresh = 32;      // segments between two reads of the global maxima once the thresholds have settled

__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return (a & b) | ((a ^ b) & c); }

__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u32 t = (u32)__shfl_xor((int)v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

__device__ __forceinline__ u32 rdlane(u32 v, u32 t) { return (u32)__builtin_amdgcn_readlane((int)v, (int)t); }

__device__ __forceinline__ u32 wave_max_u32(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u32 t = (u32)__shfl_xor((int)v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

// The work queues of the pruned kernels.  A launch's work is the sequence of (tile, batch) items, tile-major, a batch
// being a.batch consecutive segments of the table (which is sorted by the added rows).  The sequence is cut into
// eight contiguous parts, one per XCD, each handed out in order by its own ticket counter (zeroed by the host before
// the launch).  So the 512 waves of an XCD work on ~1,000 neighbouring segments of ONE tile at any time -- a dozen
// pivot genes, whose count planes and mask rows stay in that XCD's L2 -- a wave sees one or two tiles of a
// five-tile launch (five of forty: its running maxima leave LDS once per tile), and no wave waits for a slower one:
// a wave whose queue is empty takes tickets from the queue with the most work left.
// Measured alternatives: a fixed round-robin split (slowest wave 11-44 % behind the average, L2 hit rate 63 % instead
// of 84 %), batches of 16 / 64 segments (+7 % / +15 %: the window outgrows the L2), one queue per CU (+4 %), one queue
// for the chip (+8 %), an eighth of the segments in every tile per queue (+5 %; +36 % with forty tiles).
struct WorkQueue {
  // Everything the queue needs between two tickets lives in LDS (8 words per wave), read back where it is used: the
  // kernels have no scalar registers to spare across a segment.
  // [0] current queue  [1] its first item  [2] its item count  [3] batches per tile  [4] items in all  [6,7] counters
  volatile u32* st;
  __device__ __forceinline__ u32 get(int i) const { return (u32)__builtin_amdgcn_readfirstlane(st[i]); }
  __device__ __forceinline__ u32* counters() const { return (u32*)(((u64)get(7) << 32) | (u64)get(6)); }
  __device__ __forceinline__ void init(u32* counters, u32 nbatch, u32 nkt) {
    st[3] = nbatch;
    st[4] = nbatch * nkt;
    st[6] = (u32)(u64)counters;
    st[7] = (u32)((u64)counters >> 32);
  }
  __device__ __forceinline__ u32 first_of(u32 j, u32 total) const { return (u32)(((u64)total * j) >> 3); }
  __device__ __forceinline__ void select(u32 j) {
    const u32 total = get(4);
    const u32 lo = first_of(j, total);
    st[0] = j;
    st[1] = lo;
    st[2] = first_of(j + 1, total) - lo;
  }
  __device__ __forceinline__ u32 take(int lane) const {
    u32 t = 0u;
    if (lane == 0) t = atomicAdd(counters() + get(0) * kIeQueueStride, 1u);
    // reconverge HERE: without a convergent operation in the join block the compiler threads the lane != 0 edge
    // straight to the caller's loop latch, and every loop-carried scalar of the kernel turns into a vector register
    return (u32)__builtin_amdgcn_readfirstlane((int)t);
  }
  // the queue with the most tickets left (read past the caches), false when every queue is empty
  __device__ __forceinline__ bool steal(int lane) {
    const u32 total = get(4);
    u32 left = 0u;
    if (lane < 8) {
      const u32 taken = __hip_atomic_load(counters() + lane * kIeQueueStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const u32 all = first_of((u32)lane + 1u, total) - first_of((u32)lane, total);
      left = taken < all ? all - taken : 0u;
    }
    const u32 m = __builtin_amdgcn_readfirstlane(wave_max_u32(left));
    if (m == 0u) return false;
    select((u32)(__ffsll((long long)__ballot(left == m)) - 1));
    return true;
  }
};

__device__ __forceinline__ u32 xor3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96); }
__device__ __forceinline__ u32 majority(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
// borrow out of a - b - c (bitwise full subtractor): majority(~a, b, c)
__device__ __forceinline__ u32 borrow3(u32 a, u32 b, u32 c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x8E); }

// sum of 8 one-bit rows -> 4 planes (14 ops)
__device__ __forceinline__ void sum8(const u32 (&r)[8], u32 (&s)[4]) {
  const u32 a0 = xor3(r[0], r[1], r[2]), c0 = majority(r[0], r[1], r[2]);
  const u32 a1 = xor3(r[3], r[4], r[5]), c1 = majority(r[3], r[4], r[5]);
  const u32 a2 = xor3(a0, a1, r[6]), c2 = majority(a0, a1, r[6]);
  s[0] = a2 ^ r[7];
  const u32 c3 = a2 & r[7];
  const u32 b0 = xor3(c0, c1, c2), d0 = majority(c0, c1, c2);
  s[1] = b0 ^ c3;
  const u32 d1 = b0 & c3;
  s[2] = d0 ^ d1;
  s[3] = d0 & d1;
}

}  // namespace gcre
