// gcre_sparse.hip -- the sparse / bit-sliced form of the permutation (null) kernel for gfx950.
//
// Same result as k_null (gcre_kernels.hip) -- count[p][r] = sum_k popc(joined_p[k] & mask[k][r]), reference
// src/methods.h:73-88 -- computed from the other side: joined path vectors are sparse (every gene has carriers
// in <= 5 % of the patients, R/Utils.R:185-187), so instead of AND-ing all 64*W patient bits against every
// permutation, each set patient bit i of the path adds the *transposed* mask row MT[i] (one bit per
// permutation) into vertical counters:
//
//     count[p][.] = sum over set bits i of joined_p of MT[i][.]
//
// One wave owns a tile of 2048 permutations (lane l holds permutations 32*l .. 32*l+31 of the tile as one
// dword per counter plane).  Adding a row is a carry-save adder tree on full-rate bit ops (v_bitop3 / v_xor3):
// ~2.6 VALU ops per set bit per 32 permutations, against 3 issue slots (v_and + half-rate v_bcnt) per 32
// *patients* per single permutation in the dense form -- the work drops by the bit density of the path.
//
// Joins that share their paths0 row (all `count` joins of one uid, join_base.cpp:242) share its bits: the
// wave accumulates the bits of paths0[idx] once into base counters and per joined path only adds the bits
// that paths1[loc] contributes on top (path1 & ~path0).
//
// Per joined path the counters are transposed in-register (16x16 bit-matrix transpose on both halves of
// every dword) into 32 integers, looked up on the path's table diagonal and max-ed into 32 running maxima
// per lane.  Waves are independent: no barriers, per-wave LDS scratch for the bit-index list.
#include "gcre_kernels.h"

namespace gcre {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef uint16_t u16;
typedef u32 __attribute__((ext_vector_type(4))) u32x4;

constexpr int kSparseWaves = 4;           // waves per block, each fully independent
constexpr int kListFlush = 1024;          // accumulate once the list holds more than this many indices
constexpr int kListCap = kListFlush + 2048 + 16;   // + one worst-case chunk + padding

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

// Append the indices (patient numbers) of the set bits of one 64-dword chunk of a row to the LDS list; lane l
// owns dword l of the chunk.  `nb` (wave-uniform) is the list length before and after.
__device__ __forceinline__ u32 append_list(u16* list, u32 nb, u32 w, u32 chunk, int lane) {
  const u32 cnt = __builtin_popcount(w);
  const u32 incl = wave_scan_add(cnt);
  u32 pos = nb + incl - cnt;
  const u32 base = chunk * 2048u + (u32)lane * 32u;
  while (w) {
    const u32 b = __builtin_ctz(w);
    list[pos++] = (u16)(base + b);
    w &= w - 1;
  }
  return nb + __builtin_amdgcn_readlane(incl, 63);
}

// pad the list with the all-zero mask row up to a multiple of 16 and make it visible to the whole wave
__device__ __forceinline__ void seal_list(u16* list, u32 nb, int lane, u32 zrow) {
  if (lane < 16) list[nb + lane] = (u16)zrow;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Add the mask rows of `nb` listed patients into the L counter planes P (plane l = bit l of the counts).
template <int L>
__device__ __forceinline__ void accumulate(u32 (&P)[L], const u16* list, u32 nb, const u32* mt_lane, int ablate = 0) {
  if (ablate & 4) return;
  static_assert(L >= 5, "planes 0..3 are the CSA tree's ones/twos/fours/eights");
  for (u32 base = 0; base < nb; base += 16) {
    const u32x4 ia = *(const u32x4*)(list + base);        // 8 indices, broadcast read
    const u32x4 ib = *(const u32x4*)(list + base + 8);
    u32 x[16];
    if (ablate & 1) {   // diagnostics: no mask-row loads
#pragma unroll
      for (int j = 0; j < 4; j++) {
        x[2 * j] = ia[j]; x[2 * j + 1] = ia[j] * 3u; x[8 + 2 * j] = ib[j]; x[8 + 2 * j + 1] = ib[j] * 5u;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        // row i of the tile starts i*256 bytes in (< 16 MB: a 32-bit offset); one v_perm_b32 turns a packed
        // index into that offset: bytes {0, idx.lo, idx.hi, 0}
        x[2 * j] = *(const u32*)((const char*)mt_lane + __builtin_amdgcn_perm(0u, ia[j], 0x0c01000cu));
        x[2 * j + 1] = *(const u32*)((const char*)mt_lane + __builtin_amdgcn_perm(0u, ia[j], 0x0c03020cu));
        x[8 + 2 * j] = *(const u32*)((const char*)mt_lane + __builtin_amdgcn_perm(0u, ib[j], 0x0c01000cu));
        x[8 + 2 * j + 1] = *(const u32*)((const char*)mt_lane + __builtin_amdgcn_perm(0u, ib[j], 0x0c03020cu));
      }
    }
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

template <int L>
__global__ __launch_bounds__(64 * kSparseWaves) void k_null_sparse(const SparseArgs a) {
  __shared__ __attribute__((aligned(16))) u16 lists[kSparseWaves][kListCap];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Work items are (permutation tile kt, slice sl of the segment list), ordered kt-major.  Workgroups that share
  // an XCD (blockIdx % 8, MI355X_MICROARCH "Workgroup dispatch") own a contiguous range of items and walk it in
  // step, so at any time an XCD's L2 holds the transposed masks of one or two tiles only.  Pure speed: any
  // other block placement gives the same results.
  const int xcd = blockIdx.x & 7;
  const i64 wi = (i64)(blockIdx.x >> 3) * kSparseWaves + wave;       // wave index inside the XCD
  const i64 wx = a.waves_per_xcd;
  const i64 slices = 8 * wx;                                         // slices per permutation tile
  u16* list = lists[wave];
  const u32 zrow = a.mt_rows - 1;
  const int nch = (a.W32p + 63) >> 6;
  const SparseSeg* segs = (const SparseSeg*)a.segs;

  u32 nmax[32];
#pragma unroll
  for (int q = 0; q < 32; q++) nmax[q] = 0u;
  int cur_kt = -1;
  const u32* mt_lane = a.mt + lane;

  auto flush = [&]() {
    if (cur_kt < 0) return;
    u32* out = a.null_bits + (size_t)cur_kt * 2048 + lane * 32;
#pragma unroll
    for (int q = 0; q < 32; q++) {
      if (nmax[q] != 0u) atomicMax(out + q, nmax[q]);
      nmax[q] = 0u;
    }
  };

  for (int step = 0; step < a.nkt; step++) {
   const i64 item = (i64)xcd * a.nkt * wx + wi + (i64)step * wx;
   const int kt = (int)(item / slices);
   const i64 sl = item % slices;
   if (kt != cur_kt) {
     flush();
     cur_kt = kt;
     mt_lane = a.mt + (size_t)kt * a.mt_rows * 64 + lane;
   }
   for (i64 sidx = sl; sidx < a.nsegs; sidx += slices) {
    const u32 row0 = __builtin_amdgcn_readfirstlane(segs[sidx].row0);
    const u32 first = __builtin_amdgcn_readfirstlane(segs[sidx].first);
    const u32 npaths = __builtin_amdgcn_readfirstlane(segs[sidx].n);
    const u32* r0 = a.p0 + (size_t)row0 * a.S32;

    // bits of the shared paths0 row -> base counters
    u32 B[L];
#pragma unroll
    for (int l = 0; l < L; l++) B[l] = 0u;
    {
      u32 nb = 0;
      for (int c = 0; c < nch; c++) {
        const int k = c * 64 + lane;
        const u32 w = (k < a.W32p) ? r0[k] : 0u;
        nb = append_list(list, nb, w, (u32)c, lane);
        if (nb > kListFlush || c + 1 == nch) {
          seal_list(list, nb, lane, zrow);
          accumulate<L>(B, list, nb, mt_lane, a.ablate);
          nb = 0;
        }
      }
    }

    for (u32 t = 0; t < npaths; t++) {
      const u32 q = first + t;
      const u32 r1row = __builtin_amdgcn_readfirstlane(a.row1[q]) & 0x7fffffffu;
      const u32* r1 = a.p1 + (size_t)r1row * a.S32;
      u32 C[L];
#pragma unroll
      for (int l = 0; l < L; l++) C[l] = B[l];
      // only the bits paths1 adds on top of paths0
      u32 nb = 0;
      for (int c = 0; c < nch; c++) {
        const int k = c * 64 + lane;
        const u32 w = (k < a.W32p) ? (r1[k] & ~r0[k]) : 0u;
        nb = append_list(list, nb, w, (u32)c, lane);
        if (nb > kListFlush || c + 1 == nch) {
          seal_list(list, nb, lane, zrow);
          accumulate<L>(C, list, nb, mt_lane, a.ablate);
          nb = 0;
        }
      }

      // counters -> 32 integers per lane -> table diagonal -> running maxima (methods.h:96-103)
      u32 R[16];
#pragma unroll
      for (int l = 0; l < 16; l++) R[l] = (l < L) ? C[l] : 0u;
      transpose16(R);
      const u32 total = __builtin_amdgcn_readfirstlane(a.tot[q]);
      const char* diag = (const char*)((const u32*)a.t32 + sp_diag_offset(total));
#pragma unroll
      for (int j = 0; j < 16; j++) {
        u32 lo, hi;
        if (a.ablate & 2) { lo = R[j] & 0xffffu; hi = R[j] >> 16; }
        else {
          lo = *(const u32*)(diag + ((R[j] & 0xffffu) << 2));
          hi = *(const u32*)(diag + ((R[j] >> 16) << 2));
        }
        nmax[j] = (lo > nmax[j]) ? lo : nmax[j];
        nmax[j + 16] = (hi > nmax[j + 16]) ? hi : nmax[j + 16];
      }
    }
   }
  }
  flush();
}

hipError_t launch_null_sparse(const SparseArgs& a, int planes, hipStream_t stream) {
  const dim3 grid((unsigned)(8 * a.waves_per_xcd / kSparseWaves));
  const dim3 block(64 * kSparseWaves);
  if (planes <= 8) hipLaunchKernelGGL(k_null_sparse<8>, grid, block, 0, stream, a);
  else if (planes <= 10) hipLaunchKernelGGL(k_null_sparse<10>, grid, block, 0, stream, a);
  else if (planes <= 12) hipLaunchKernelGGL(k_null_sparse<12>, grid, block, 0, stream, a);
  else hipLaunchKernelGGL(k_null_sparse<16>, grid, block, 0, stream, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// transposed masks: mt[kt][i][lane] bit q = mask bit of patient i under permutation kt*2048 + lane*32 + q
// ------------------------------------------------------------------------------------------------
__global__ void k_build_mt(const u32* masks, int W32p, int Kpad, int nkt, u32 mt_rows, u32* mt) {
  // one thread per (dword row k32 of the masks, tile, lane): a 32x32 bit transpose
  const i64 total = (i64)W32p * nkt * 64;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const int kt = (int)((i >> 6) % nkt);
    const int k32 = (int)((i >> 6) / nkt);
    u32 m[32];
#pragma unroll
    for (int q = 0; q < 32; q++) {
      const i64 col = (i64)kt * 2048 + lane * 32 + q;
      m[q] = (col < Kpad) ? masks[(size_t)k32 * Kpad + col] : 0u;
    }
    for (int b = 0; b < 32; b++) {
      u32 v = 0;
#pragma unroll
      for (int q = 0; q < 32; q++) v |= ((m[q] >> b) & 1u) << q;
      mt[((size_t)kt * mt_rows + (size_t)k32 * 32 + b) * 64 + lane] = v;
    }
  }
}

hipError_t launch_build_mt(const uint32_t* masks, int W32p, int Kpad, int nkt, uint32_t mt_rows, uint32_t* mt,
                           hipStream_t stream) {
  const i64 total = (i64)W32p * nkt * 64;
  if (total == 0) return hipSuccess;
  const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL(k_build_mt, dim3(grid), dim3(256), 0, stream, masks, W32p, Kpad, nkt, mt_rows, mt);
  return hipGetLastError();
}

}  // namespace gcre
