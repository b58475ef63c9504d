// gcre_bitslice.h -- bit-sliced ("vertical") counter arithmetic shared by the sparse null kernels.
// Lane l of a wave holds permutations 32*l .. 32*l+31 of a 2048-permutation tile as one dword per counter plane
// (plane p = bit p of the 32 counts).  Everything here is full-rate bit logic (v_bitop3 / v_xor3).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gcre {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef u32 __attribute__((ext_vector_type(4))) u32x4;
typedef u32 __attribute__((ext_vector_type(16))) u32x16;

#define GCRE_CONSTANT __attribute__((address_space(4)))

__device__ __forceinline__ u32 sp_diag_offset(u32 t) { return (u32)(((u64)t * (u64)(t + 1)) >> 1); }

// carry-save adder: a + b + c = 2*hi + lo, bitwise over 32 permutations
__device__ __forceinline__ void csa(u32& hi, u32& lo, u32 a, u32 b, u32 c) {
  const u32 u = a ^ b;
  hi = (a & b) | (u & c);
  lo = u ^ c;
}

// inclusive prefix sum over the 64 lanes on the DPP crossbar (no LDS traffic): row_shr 1/2/3, then 4 and 8 inside
// each row of 16, then row_bcast:15 / row_bcast:31 carry the row totals forward
__device__ __forceinline__ u32 wave_scan_add(u32 v) {
  u32 s = v;
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);   // row_shr:3
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, true);   // row_shr:4, banks 1-3
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, true);   // row_shr:8, banks 2-3
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, true);   // row_bcast:15 -> rows 1, 3
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, true);   // row_bcast:31 -> rows 2, 3
  return s;
}

// Add 16 mask rows (one dword per lane each) into the L counter planes P (plane l = bit l of the counts).
template <int L>
__device__ __forceinline__ void add16(u32 (&P)[L], const u32 (&x)[16]) {
  static_assert(L >= 5, "planes 0..3 are the CSA tree's ones/twos/fours/eights");
  u32 t0, t1, t2, t3, f0, f1, e0, e1, s;
  csa(t0, P[0], P[0], x[0], x[1]);
  csa(t1, P[0], P[0], x[2], x[3]);
  csa(f0, P[1], P[1], t0, t1);
  csa(t2, P[0], P[0], x[4], x[5]);
  csa(t3, P[0], P[0], x[6], x[7]);
  csa(f1, P[1], P[1], t2, t3);
  csa(e0, P[2], P[2], f0, f1);
  csa(t0, P[0], P[0], x[8], x[9]);
  csa(t1, P[0], P[0], x[10], x[11]);
  csa(f0, P[1], P[1], t0, t1);
  csa(t2, P[0], P[0], x[12], x[13]);
  csa(t3, P[0], P[0], x[14], x[15]);
  csa(f1, P[1], P[1], t2, t3);
  csa(e1, P[2], P[2], f0, f1);
  csa(s, P[3], P[3], e0, e1);
  // ripple the weight-16 carry through the remaining planes
#pragma unroll
  for (int l = 4; l < L; l++) {
    const u32 c = P[l] & s;
    P[l] ^= s;
    s = c;
  }
}

// Add 4 mask rows: the tail of a list (lists are padded to 4 entries, so at most 3 loads are wasted per list).
template <int L>
__device__ __forceinline__ void add4(u32 (&P)[L], const u32 (&x)[4]) {
  u32 t0, t1, s;
  csa(t0, P[0], P[0], x[0], x[1]);
  csa(t1, P[0], P[0], x[2], x[3]);
  csa(s, P[1], P[1], t0, t1);
#pragma unroll
  for (int l = 2; l < L; l++) {
    const u32 c = P[l] & s;
    P[l] ^= s;
    s = c;
  }
}

// 16x16 bit-matrix transpose of the low and of the high 16 bits of R[0..15] at once:
// afterwards bit l of the low (high) half of R[q] is the former bit q (q + 16) of R[l].
__device__ __forceinline__ void transpose16(u32 (&R)[16]) {
#define GCRE_TSTAGE(S, MASK)                                   \
  _Pragma("unroll") for (int i = 0; i < 16; i++) {             \
    if ((i & (S)) == 0) {                                      \
      const u32 a = R[i], b = R[i + (S)];                      \
      R[i] = (a & (MASK)) | ((b << (S)) & ~(MASK));            \
      R[i + (S)] = ((a >> (S)) & (MASK)) | (b & ~(MASK));      \
    }                                                          \
  }
  GCRE_TSTAGE(8, 0x00ff00ffu)
  GCRE_TSTAGE(4, 0x0f0f0f0fu)
  GCRE_TSTAGE(2, 0x33333333u)
  GCRE_TSTAGE(1, 0x55555555u)
#undef GCRE_TSTAGE
}

// 16 mask rows addressed by 16 wave-uniform byte offsets (SGPRs): lane*4 is the vector offset
__device__ __forceinline__ void load16(u32 (&x)[16], __amdgpu_buffer_rsrc_t mt, u32 lane4, const u32x16 offs) {
#pragma unroll
  for (int j = 0; j < 16; j++) x[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
}

}  // namespace gcre
