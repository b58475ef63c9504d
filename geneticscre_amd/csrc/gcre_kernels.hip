// gcre_kernels.hip -- gfx950 (CDNA4) kernels of the permutation-tested path-join scorer.
//
// The hot kernel is k_null: for a tile of joined paths x a tile of label permutations it computes
//     count[p][r] = sum_k popcount((paths0[idx_p][k] | paths1[loc_p][k]) & mask[k][r])
// -- the inner loops of JoinMethod1/2::score_permute (reference src/methods.h:73-88, 162-210) -- looks
// the score up on the path's table diagonal and keeps the per-permutation running maximum
// (methods.h:96-103, 220-230).  Integer bit work: v_and_b32 + v_bcnt_u32_b32, no MFMA.
//
// Mapping onto the hardware (wave = 64 lanes):
//   * lanes own permutations (R per lane): the running maximum never leaves the lane;
//   * joined-path words are wave-uniform: fetched with scalar loads (s_load_dwordx8) from HBM/L2 into
//     SGPRs, OR-ed on the scalar unit, and fed to the VALU as the scalar operand of v_and_b32;
//   * the permutation-mask tile [8 dwords][64*R perms] is staged through LDS (double buffered, one
//     barrier per chunk) and shared by the 4 waves of a block; each wave keeps its 8*R mask registers
//     for TPW paths, so one LDS read feeds 2*TPW VALU ops;
//   * a block stays on one permutation tile for all its path tiles: 64*R atomics per block at the end.
#include "gcre_kernels.h"

namespace gcre {

static inline int64_t hmin(int64_t a, int64_t b) { return a < b ? a : b; }

typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef u32 __attribute__((ext_vector_type(2))) u32x2;
typedef u32 __attribute__((ext_vector_type(4))) u32x4;
typedef u32 __attribute__((ext_vector_type(8))) u32x8;

#define GCRE_CONSTANT __attribute__((address_space(4)))

// Read-only kernel inputs are addressed through the constant address space so that wave-uniform
// addresses select scalar loads.
template <typename T>
__device__ __forceinline__ const T GCRE_CONSTANT* as_const(const T* p) {
  return (const T GCRE_CONSTANT*)p;
}

__device__ __forceinline__ u32 diag_offset(u32 t) { return (u32)(((u64)t * (u64)(t + 1)) >> 1); }

// ------------------------------------------------------------------------------------------------
// k_null
// ------------------------------------------------------------------------------------------------
template <int R>
__device__ __forceinline__ void read_mask_row(const u32* lds_row, int lane, u32 (&m)[R]) {
  if constexpr (R == 1) {
    m[0] = lds_row[lane];
  } else if constexpr (R == 2) {
    u32x2 v = *(const u32x2*)(lds_row + lane * 2);
    m[0] = v.x; m[1] = v.y;
  } else if constexpr (R == 4) {
    u32x4 v = *(const u32x4*)(lds_row + lane * 4);
    m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w;
  } else {
    static_assert(R == 8, "R in {1,2,4,8}");
    // lane owns permutations {4*lane .. 4*lane+3} and {256 + 4*lane ..}: both ds_read_b128 have a 16-byte lane
    // stride (a 32-byte stride is a 2-way bank conflict for b128 reads)
    u32x4 v = *(const u32x4*)(lds_row + lane * 4);
    u32x4 w = *(const u32x4*)(lds_row + 256 + lane * 4);
    m[0] = v.x; m[1] = v.y; m[2] = v.z; m[3] = v.w;
    m[4] = w.x; m[5] = w.y; m[6] = w.z; m[7] = w.w;
  }
}

template <int WC> struct RowChunk;
template <> struct RowChunk<4> { typedef u32x4 type; };
template <> struct RowChunk<8> { typedef u32x8 type; };

// M method, R permutations per lane, TPW joined paths per wave per tile, WC dwords of every row per
// chunk (one s_load_dwordx4 / x8), OCC waves per SIMD the register allocation must admit
template <int M, int R, int TPW, int WC, int OCC>
__global__ __launch_bounds__(kNullBlock, OCC) void k_null(const NullArgs a) {
  typedef typename RowChunk<WC>::type rowv;
  constexpr int NW = kNullBlock / 64;   // waves per block
  constexpr int PT = 64 * R;            // permutations per tile
  constexpr int TPB = NW * TPW;         // joined paths per block tile
  constexpr int CHUNK = WC * PT;        // dwords per staged mask chunk
  constexpr int VEC = CHUNK / 4 / kNullBlock;   // uint4 per thread per chunk (R=4: 2)
  static_assert(CHUNK % (4 * kNullBlock) == 0 || CHUNK < 4 * kNullBlock, "staging shape");

  __shared__ __attribute__((aligned(16))) u32 lds[2][CHUNK];
  __shared__ u32 red[NW][PT];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kt = blockIdx.x % a.nkt;
  const int g = blockIdx.x / a.nkt;

  const u32 GCRE_CONSTANT* P0 = as_const(a.p0);
  const u32 GCRE_CONSTANT* P1 = as_const(a.p1);
  const u32 GCRE_CONSTANT* ROW0 = as_const(a.row0);
  const u32 GCRE_CONSTANT* ROW1 = as_const(a.row1);
  const u32 GCRE_CONSTANT* TOT = as_const(a.tot);

  const int nchunks = a.W32p / WC;
  const u32* mask_tile = a.masks + (size_t)kt * PT;   // column offset of this permutation tile

  // ---- mask staging: chunk c = rows [c*WC, c*WC+WC) x PT columns, row-major in LDS ----
  // thread e of the block moves uint4 number e (+ i*256) of the chunk; its byte offset from the chunk's
  // first row is loop-invariant and fits 32 bits, the chunk base is wave-uniform
  constexpr int NV = (VEC > 0) ? VEC : 1;
  u32x4 stage[NV];
  u32 voff[NV];
#pragma unroll
  for (int i = 0; i < NV; i++) {
    const int e = tid + i * kNullBlock;
    voff[i] = (u32)((e / (PT / 4)) * a.Kpad + (e % (PT / 4)) * 4) * 4u;
  }
  const size_t chunk_bytes = (size_t)WC * a.Kpad * 4;
  auto stage_load = [&](int c) {
    const char* cb = (const char*)mask_tile + (size_t)c * chunk_bytes;
#pragma unroll
    for (int i = 0; i < NV; i++)
      if (VEC > 0 || tid + i * kNullBlock < CHUNK / 4) stage[i] = *(const u32x4*)(cb + voff[i]);
  };
  auto stage_store = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NV; i++) {
      const int e = tid + i * kNullBlock;
      if (VEC > 0 || e < CHUNK / 4) *(u32x4*)(&lds[buf][e * 4]) = stage[i];
    }
  };

  u32 nmax[R];
#pragma unroll
  for (int j = 0; j < R; j++) nmax[j] = 0u;

  stage_load(0);
  stage_store(0);
  __syncthreads();
  int buf = 0;

  const u32 s8 = (u32)(a.S32 >> 3);   // row stride in 8-dword units (rows are 32-byte aligned)
  const u32 h8 = (u32)(a.W32p >> 3);  // (+)/(-) half stride in the same units

  for (i64 pt = g; pt < a.npt; pt += a.pgroups) {
    const i64 qbase = pt * TPB + (i64)wave * TPW;
    u32 acc[M][TPW][R];
#pragma unroll
    for (int h = 0; h < M; h++)
#pragma unroll
      for (int t = 0; t < TPW; t++)
#pragma unroll
        for (int j = 0; j < R; j++) acc[h][t][j] = 0u;

    // row offsets of this wave's TPW joined paths, in 32-byte units, wave-uniform (SGPRs)
    u32 o0[TPW], o1[TPW], o1n[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      const u32 r0 = ROW0[qbase + t];
      const u32 r1raw = ROW1[qbase + t];
      o0[t] = r0 * s8;
      o1[t] = (r1raw & 0x7fffffffu) * s8;
      o1n[t] = o1[t];
      if constexpr (M == 2) {
        // (+) half of path1 is its second half when the relation flips the sign (methods.h:140-142)
        const u32 swap = r1raw >> 31;
        o1n[t] = o1[t] + (swap ? 0u : h8);
        o1[t] = o1[t] + (swap ? h8 : 0u);
      }
    }

    // software pipeline over the flattened (chunk, path) sequence: the scalar loads of the next
    // path's words are in flight while the VALU works on the current one
    rowv nx[M], ny[M];
    auto fetch = [&](int c, int t) {
      const u32 GCRE_CONSTANT* b0 = P0 + (size_t)c * WC;
      const u32 GCRE_CONSTANT* b1 = P1 + (size_t)c * WC;
      nx[0] = *(const rowv GCRE_CONSTANT*)(b0 + ((size_t)o0[t] << 3));
      ny[0] = *(const rowv GCRE_CONSTANT*)(b1 + ((size_t)o1[t] << 3));
      if constexpr (M == 2) {
        nx[M - 1] = *(const rowv GCRE_CONSTANT*)(b0 + ((size_t)(o0[t] + h8) << 3));
        ny[M - 1] = *(const rowv GCRE_CONSTANT*)(b1 + ((size_t)o1n[t] << 3));
      }
    };
    fetch(0, 0);

    for (int c = 0; c < nchunks; c++) {
      // prefetch the next chunk of this block's (periodic) mask stream
      const int cn = (c + 1 == nchunks) ? 0 : c + 1;
      stage_load(cn);

      u32 m[WC][R];
#pragma unroll
      for (int w = 0; w < WC; w++) read_mask_row<R>(&lds[buf][w * PT], lane, m[w]);

#pragma unroll
      for (int t = 0; t < TPW; t++) {
        rowv jn[M];
#pragma unroll
        for (int h = 0; h < M; h++)
#pragma unroll
          for (int w = 0; w < WC; w++)   // OR on the scalar unit; the VALU only sees the joined word
            jn[h][w] = __builtin_amdgcn_readfirstlane(nx[h][w] | ny[h][w]);
        if (t + 1 < TPW) fetch(c, t + 1);
        else fetch(cn, 0);
#pragma unroll
        for (int h = 0; h < M; h++)
#pragma unroll
          for (int w = 0; w < WC; w++)
#pragma unroll
            for (int j = 0; j < R; j++) acc[h][t][j] += __builtin_popcount(jn[h][w] & m[w][j]);
        __builtin_amdgcn_sched_barrier(0);   // keep each path's scalar loads / ORs in its own region
      }

      stage_store(buf ^ 1);
      __syncthreads();
      buf ^= 1;
    }

    // ---- table lookup on each path's diagonal + running maximum (methods.h:96-103 / 220-230) ----
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      if (qbase + t < a.npaths) {
        if constexpr (M == 1) {
          const u32 total = TOT[qbase + t];
          // wave-uniform diagonal base + 32-bit per-lane byte offset (keeps the counters 32-bit registers)
          const char* diag = (const char*)((const u32*)a.t32 + diag_offset(total));
#pragma unroll
          for (int j = 0; j < R; j++) {
            const u32 v = *(const u32*)(diag + (acc[0][t][j] << 2));
            nmax[j] = (v > nmax[j]) ? v : nmax[j];
          }
        } else {
          const u32 tp = TOT[2 * (qbase + t)];
          const u32 tn = TOT[2 * (qbase + t) + 1];
          const char* dp = (const char*)(a.d64 + diag_offset(tp));
          const char* dn = (const char*)(a.d64 + diag_offset(tn));
#pragma unroll
          for (int j = 0; j < R; j++) {
            const double s = *(const double*)(dp + (acc[0][t][j] << 3)) + *(const double*)(dn + (acc[M - 1][t][j] << 3));
            float f = (float)s;
            f = (f > 0.0f) ? f : 0.0f;       // the running maximum starts at 0; NaN and negatives never win
            const u32 v = __float_as_uint(f);
            nmax[j] = (v > nmax[j]) ? v : nmax[j];
          }
        }
      }
    }
  }

  // ---- block reduction, then one atomic per permutation ----
#pragma unroll
  for (int j = 0; j < R; j++) {
    // register j of a lane <-> column of the permutation tile (see read_mask_row)
    const int col = (R == 8) ? ((j >> 2) * 256 + lane * 4 + (j & 3)) : (lane * R + j);
    red[wave][col] = nmax[j];
  }
  __syncthreads();
  for (int i = tid; i < PT; i += kNullBlock) {
    u32 v = red[0][i];
#pragma unroll
    for (int w = 1; w < NW; w++) v = (red[w][i] > v) ? red[w][i] : v;
    if (v != 0u) atomicMax(a.null_bits + (size_t)kt * PT + i, v);
  }
}

NullConfig null_config(int method, int K) {
  NullConfig c;
  c.R = (K <= 64) ? 1 : (K <= 128) ? 2 : (K <= 256) ? 4 : 8;
  c.TPW = 64 / c.R / method;
  if (c.TPW > 16) c.TPW = 16;
  c.perm_tile = 64 * c.R;
  c.path_tile = (kNullBlock / 64) * c.TPW;
  return c;
}

template <int M, int R, int TPW, int WC, int OCC>
static hipError_t launch_null_t(const NullArgs& a, hipStream_t stream) {
  const dim3 grid((unsigned)(a.nkt * a.pgroups));
  hipLaunchKernelGGL((k_null<M, R, TPW, WC, OCC>), grid, dim3(kNullBlock), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_null(const NullArgs& a, int method, const NullConfig& cfg, hipStream_t stream) {
  if (method == 1) {
    switch (cfg.R) {
      case 1: return launch_null_t<1, 1, 16, 8, 4>(a, stream);
      case 2: return launch_null_t<1, 2, 16, 8, 4>(a, stream);
      case 4: return launch_null_t<1, 4, 16, 8, 2>(a, stream);
      default: return launch_null_t<1, 8, 8, 4, 4>(a, stream);
    }
  }
  switch (cfg.R) {
    case 1: return launch_null_t<2, 1, 16, 8, 4>(a, stream);
    case 2: return launch_null_t<2, 2, 16, 8, 2>(a, stream);
    case 4: return launch_null_t<2, 4, 8, 8, 2>(a, stream);
    default: return launch_null_t<2, 8, 4, 4, 3>(a, stream);
  }
}

// ------------------------------------------------------------------------------------------------
// input packing
// ------------------------------------------------------------------------------------------------

// PathSet::load (gcre_paths.h:56-70): bit c of row r set iff data[r][c] != 0, (+) half only.
__global__ void k_pack_dense(const int32_t* data, i64 nrow, int ncol, int col_major, u64* rows, int S) {
  const int words = (ncol + 63) / 64;
  const i64 total = nrow * words;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const i64 r = i / words;
    const int k = (int)(i % words);
    u64 v = 0;
    const int c1 = min(ncol, (k + 1) * 64);
    for (int c = k * 64; c < c1; c++) {
      const int32_t d = col_major ? data[(size_t)c * nrow + r] : data[(size_t)r * ncol + c];
      if (d != 0) v |= (u64)1 << (c & 63);
    }
    rows[(size_t)r * S + k] = v;
  }
}

hipError_t launch_pack_dense(const int32_t* data, int64_t nrow, int ncol, int col_major, uint64_t* rows, int S,
                             hipStream_t stream) {
  const i64 total = nrow * ((ncol + 63) / 64);
  if (total == 0) return hipSuccess;
  const int grid = (int)hmin((total + 255) / 256, 8192);
  hipLaunchKernelGGL(k_pack_dense, dim3(grid), dim3(256), 0, stream, data, nrow, ncol, col_major, rows, S);
  return hipGetLastError();
}

// PathSet::select (gcre_paths.h:82-92): row gather.
__global__ void k_select(const u64* from, const int32_t* idx, i64 n, int S, u64* out) {
  const i64 total = n * S;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const i64 r = i / S;
    const int k = (int)(i % S);
    out[i] = from[(size_t)idx[r] * S + k];
  }
}

hipError_t launch_select(const uint64_t* from, const int32_t* idx, int64_t n, int S, uint64_t* out, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const int grid = (int)hmin((n * S + 255) / 256, 16384);
  hipLaunchKernelGGL(k_select, dim3(grid), dim3(256), 0, stream, from, idx, n, S, out);
  return hipGetLastError();
}

// setPermutedCases (join_base.cpp:85-125): mask_r = case_mask XOR flipped_r; rows beyond the supplied
// ones reuse row r % nrows_in; stored word-major / permutation-minor.
__global__ void k_masks_from_ints(const int32_t* perms, int nrows_in, int ncol, int col_major, int n_cases, int K,
                                  int W32p, int Kpad, u32* masks) {
  const i64 total = (i64)W32p * K;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int k = (int)(i / K);
    const int r = (int)(i % K);
    const int s = r % nrows_in;
    u32 v = 0;
    const int c1 = min(ncol, (k + 1) * 32);
    for (int c = k * 32; c < c1; c++) {
      const int32_t d = col_major ? perms[(size_t)c * nrows_in + s] : perms[(size_t)s * ncol + c];
      const u32 is_case = (c < n_cases) ? 1u : 0u;
      const u32 flipped = (d != 1) ? 1u : 0u;
      v |= (is_case ^ flipped) << (c & 31);
    }
    masks[(size_t)k * Kpad + r] = v;
  }
}

hipError_t launch_masks_from_ints(const int32_t* perms, int nrows_in, int ncol, int col_major, const Geometry& g,
                                  uint32_t* masks, hipStream_t stream) {
  const i64 total = (i64)(2 * g.Wp) * g.K;
  if (total == 0) return hipSuccess;
  const int grid = (int)hmin((total + 255) / 256, 16384);
  hipLaunchKernelGGL(k_masks_from_ints, dim3(grid), dim3(256), 0, stream, perms, nrows_in, ncol, col_major, g.n_cases,
                     g.K, 2 * g.Wp, g.Kpad, masks);
  return hipGetLastError();
}

__global__ void k_masks_from_words(const u64* packed, int nrows_in, int W, int K, int W32p, int Kpad, u32* masks) {
  const i64 total = (i64)W32p * K;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (i64)gridDim.x * blockDim.x) {
    const int k = (int)(i / K);
    const int r = (int)(i % K);
    const int s = r % nrows_in;
    u32 v = 0;
    if ((k >> 1) < W) {
      const u64 w = packed[(size_t)s * W + (k >> 1)];
      v = (k & 1) ? (u32)(w >> 32) : (u32)w;
    }
    masks[(size_t)k * Kpad + r] = v;
  }
}

hipError_t launch_masks_from_words(const uint64_t* packed, int nrows_in, const Geometry& g, uint32_t* masks,
                                   hipStream_t stream) {
  const i64 total = (i64)(2 * g.Wp) * g.K;
  if (total == 0) return hipSuccess;
  const int grid = (int)hmin((total + 255) / 256, 16384);
  hipLaunchKernelGGL(k_masks_from_words, dim3(grid), dim3(256), 0, stream, packed, nrows_in, g.W, g.K, 2 * g.Wp,
                     g.Kpad, masks);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// joined-path ordinal -> rows (the two nested loops of JoinExec::join, join_base.cpp:230-250)
// ------------------------------------------------------------------------------------------------
constexpr int kExpandRun = 8;   // consecutive joined paths per thread: one binary search, then a walk along the uids

__global__ void k_expand(const i64* path_idx, const i64* location, i64 n_uids, const int32_t* signs, int path_length,
                         int method, i64 first, i64 count, u32* row0, u32* row1) {
  const i64 runs = (count + kExpandRun - 1) / kExpandRun;
  for (i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x; r < runs; r += (i64)gridDim.x * blockDim.x) {
    const i64 i0 = r * kExpandRun;
    const i64 i1 = i0 + kExpandRun < count ? i0 + kExpandRun : count;
    // last uid whose first path ordinal is <= p (uids with count 0 share an ordinal with their successor)
    i64 lo = 0, hi = n_uids;   // invariant: path_idx[lo] <= p < path_idx[hi]
    {
      const i64 p = first + i0;
      while (hi - lo > 1) {
        const i64 mid = (lo + hi) >> 1;
        if (path_idx[mid] <= p) lo = mid; else hi = mid;
      }
    }
    i64 idx = lo;
    i64 next = path_idx[idx + 1];   // first ordinal of the uid after idx
    for (i64 i = i0; i < i1; i++) {
      const i64 p = first + i;
      while (next <= p) {           // the same "last uid with path_idx <= p" as the search: empty uids are stepped over
        idx++;
        next = path_idx[idx + 1];
      }
      const i64 loc = location[idx] + (p - path_idx[idx]);
      u32 swap = 0;
      if (method == 2) {
        // UidRelSet::need_flip (gcre.h:71-81): sign == 1 keeps path1's halves, otherwise they swap
        int sign;
        if (path_length > 3) sign = signs[idx];
        else if (path_length < 3) sign = signs[loc];
        else sign = (signs[idx] + signs[loc] == 0) ? -1 : 1;
        swap = (sign == 1) ? 0u : 1u;
      }
      row0[i] = (u32)idx;
      row1[i] = (u32)loc | (swap << 31);
    }
  }
}

hipError_t launch_expand(const int64_t* path_idx, const int64_t* location, int64_t n_uids, const int32_t* signs,
                         int path_length, int method, int64_t first, int64_t count, uint32_t* row0, uint32_t* row1,
                         hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const int grid = (int)hmin((count + 255) / 256, 16384);
  hipLaunchKernelGGL(k_expand, dim3(grid), dim3(256), 0, stream, path_idx, location, n_uids, signs, path_length, method,
                     first, count, row0, row1);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// real-label statistics, observed score and kept rows: one wave per joined path
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// order-preserving image of a double; 0 is reserved for "not a candidate" (score not > -inf, or NaN:
// the reference only inserts when score > heap minimum, which starts at -inf -- methods.h:91)
__device__ __forceinline__ u64 score_key(double s) {
  if (!(s > -__builtin_inf())) return 0;
  const u64 b = (u64)__double_as_longlong(s);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

template <int M>
__global__ __launch_bounds__(256) void k_stats(const StatsArgs a) {
  const int lane = threadIdx.x & 63;
  const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
  const int Wp = a.Wp;
  u32 wave_max_tot = 0, wave_modes = 0;
  for (i64 i = wave; i < a.count; i += nwaves) {
    const u32 r0 = a.row0[i];
    const u32 r1raw = a.row1[i];
    const u32 r1 = r1raw & 0x7fffffffu;
    const u64* x = a.p0 + (size_t)r0 * a.S;
    const u64* y = a.p1 + (size_t)r1 * a.S;
    // row of the reduced operand (inclusion-exclusion form); bit 31 of an index entry swaps its halves once more
    const u32 zraw = a.zindex ? (u32)a.zindex[r1] : r1;
    const u32 rz = zraw & 0x7fffffffu;
    const u32 zflip = (r1raw ^ (a.zindex ? zraw : 0u)) & 0x80000000u;
    u64* out = a.res ? a.res + (size_t)(a.first + i) * a.S : nullptr;
    if constexpr (M == 1) {
      u32 cs = 0, ct = 0, dl = 0, ov = 0, px = 0, extra = 0;
      const u64* z = a.pz ? a.pz + (size_t)rz * a.S : y;
      for (int k = lane; k < Wp; k += 64) {
        const u64 xk = x[k], yk = y[k];
        const u64 j = xk | yk;
        const u64 cm = a.case_mask[k];
        cs += __popcll(j & cm);      // methods.h:77-78
        ct += __popcll(j & ~cm);
        if (a.pz) {
          const u64 zk = z[k];
          dl += __popcll(zk & ~xk);     // bits the reduced row adds on top of paths0 ...
          ov += __popcll(zk & xk);      // ... and the bits it shares with it
          px += __popcll(xk);
          extra += __popcll(zk & ~j);
        } else {
          dl += __popcll(yk & ~xk);     // bits paths1 adds on top of paths0 (sparse kernel's per-path list length)
        }
        if (out) out[k] = j;
      }
      cs = wave_sum(cs);
      ct = wave_sum(ct);
      if (a.dcnt) {
        dl = wave_sum(dl);
        if (a.pz) {
          ov = wave_sum(ov);
          px = wave_sum(px);
          extra = wave_sum(extra);
          if (lane == 0) {
            const u32 mode = (a.ie_bias >= 0 && ov + (u32)a.ie_bias < dl) ? 1u : 0u;
            a.dcnt[i] = max(8u, ((mode ? ov : dl) + 7u) & ~7u) | mode;   // the IE kernel always loads 8 entries
            a.rowz[i] = rz | zflip;
            // paths0 | reduced row must be the joined path: reduced row inside it, same number of carriers
            if (extra != 0u || px + dl != cs + ct) *a.bad = 1u;
            wave_modes += mode;
          }
        } else if (lane == 0) {
          a.dcnt[i] = (dl + 3u) & ~3u;
        }
      }
      if (lane == 0) {
        const u32 total = cs + ct;
        const double s = a.dvt[(size_t)diag_offset(total) + cs];   // vt[cases][ctrls], methods.h:90
        a.key[i] = score_key(s);
        a.tot[i] = total;
        a.cases[i] = cs;
        a.ctrls[i] = ct;
        wave_max_tot = max(wave_max_tot, total);
      }
    } else {
      const bool swap = (r1raw >> 31) != 0;
      const u64* yp = swap ? y + Wp : y;
      const u64* yn = swap ? y : y + Wp;
      u32 case_pos = 0, ctrl_neg = 0, case_neg = 0, ctrl_pos = 0, dlp = 0, dln = 0;
      u32 ovp = 0, ovn = 0, pxp = 0, pxn = 0, extra = 0;
      const u64* z = a.pz ? a.pz + (size_t)rz * a.S : y;
      const u64* zp = zflip ? z + Wp : z;
      const u64* zn = zflip ? z : z + Wp;
      for (int k = lane; k < Wp; k += 64) {
        const u64 xp = x[k], xn = x[Wp + k];
        const u64 bp = xp | yp[k];              // methods.h:164-165
        const u64 bn = xn | yn[k];
        const u64 cm = a.case_mask[k];
        dlp += __popcll(zp[k] & ~xp);           // bits the joined row adds to the (+) / (-) half
        dln += __popcll(zn[k] & ~xn);
        if (a.pz) {
          ovp += __popcll(zp[k] & xp);
          ovn += __popcll(zn[k] & xn);
          pxp += __popcll(xp);
          pxn += __popcll(xn);
          extra += __popcll(zp[k] & ~bp) + __popcll(zn[k] & ~bn);
        }
        case_pos += __popcll(bp & cm);          // methods.h:182-185
        ctrl_neg += __popcll(bp & ~cm);
        case_neg += __popcll(bn & ~cm);
        ctrl_pos += __popcll(bn & cm);
        if (out) { out[k] = bp; out[Wp + k] = bn; }
      }
      case_pos = wave_sum(case_pos);
      ctrl_neg = wave_sum(ctrl_neg);
      case_neg = wave_sum(case_neg);
      ctrl_pos = wave_sum(ctrl_pos);
      if (a.dcnt) {
        dlp = wave_sum(dlp);
        dln = wave_sum(dln);
        if (a.pz) {
          ovp = wave_sum(ovp);
          ovn = wave_sum(ovn);
          pxp = wave_sum(pxp);
          pxn = wave_sum(pxn);
          extra = wave_sum(extra);
          if (lane == 0) {
            const u32 mp = (a.ie_bias >= 0 && ovp + (u32)a.ie_bias < dlp) ? 1u : 0u;
            const u32 mn = (a.ie_bias >= 0 && ovn + (u32)a.ie_bias < dln) ? 1u : 0u;
            a.dcnt[2 * i] = max(8u, ((mp ? ovp : dlp) + 7u) & ~7u) | mp;
            a.dcnt[2 * i + 1] = max(8u, ((mn ? ovn : dln) + 7u) & ~7u) | mn;
            a.rowz[i] = rz | zflip;
            if (extra != 0u || pxp + dlp != case_pos + ctrl_neg || pxn + dln != case_neg + ctrl_pos) *a.bad = 1u;
            wave_modes += mp + mn;
          }
        } else if (lane == 0) {
          a.dcnt[2 * i] = (dlp + 3u) & ~3u;
          a.dcnt[2 * i + 1] = (dln + 3u) & ~3u;
        }
      }
      if (lane == 0) {
        const u32 tp = case_pos + ctrl_neg, tn = case_neg + ctrl_pos;
        // vt[case_pos][ctrl_neg] + vt[case_neg][ctrl_pos], methods.h:255
        const double s = a.dvt[(size_t)diag_offset(tp) + case_pos] + a.dvt[(size_t)diag_offset(tn) + case_neg];
        a.key[i] = score_key(s);
        a.tot[2 * i] = tp;
        a.tot[2 * i + 1] = tn;
        a.cases[i] = case_pos + case_neg;       // methods.h:256-257
        a.ctrls[i] = ctrl_pos + ctrl_neg;
        wave_max_tot = max(wave_max_tot, max(tp, tn));
      }
    }
  }
  if (a.max_tot && lane == 0 && wave_max_tot) atomicMax(a.max_tot, wave_max_tot);
  if (a.bad && lane == 0 && wave_modes) atomicAdd(a.bad + 1, wave_modes);   // statistics: overlap-mode lists
}

hipError_t launch_stats(const StatsArgs& a, int method, hipStream_t stream) {
  if (a.count == 0) return hipSuccess;
  const int grid = (int)hmin((a.count + 3) / 4, 256 * 16);
  if (method == 1) hipLaunchKernelGGL(k_stats<1>, dim3(grid), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(k_stats<2>, dim3(grid), dim3(256), 0, stream, a);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// top-k selection: MSB-first radix select on the score keys, then an index-ordered tie cut
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hist(const u64* key, i64 count, int shift, u64 prefix, u32* hist256) {
  __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (i64)gridDim.x * blockDim.x) {
    const u64 k = key[i];
    // keys whose higher bytes equal the prefix chosen so far (shift == 56: every key)
    const bool match = (shift == 56) ? true : ((k >> (shift + 8)) == prefix);
    if (match) atomicAdd(&h[(u32)(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist256[threadIdx.x], h[threadIdx.x]);
}

hipError_t launch_hist(const uint64_t* key, int64_t count, int shift, uint64_t prefix, uint32_t* hist256,
                       hipStream_t stream) {
  const int grid = (int)hmin((count + 2047) / 2048, 2048);
  hipLaunchKernelGGL(k_hist, dim3(grid > 0 ? grid : 1), dim3(256), 0, stream, key, count, shift, prefix, hist256);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_select_init(SelectState* st, i64 need, u32* hist256) {
  hist256[threadIdx.x] = 0;
  if (threadIdx.x == 0) *st = SelectState{0, need, 0, 0, 0};
}

__global__ __launch_bounds__(256) void k_hist_st(const u64* key, i64 count, int shift, const SelectState* st, u32* hist256) {
  __shared__ u32 h[256];
  h[threadIdx.x] = 0;
  __syncthreads();
  const u64 prefix = st->prefix;
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (i64)gridDim.x * blockDim.x) {
    const u64 k = key[i];
    const bool match = (shift == 56) ? true : ((k >> (shift + 8)) == prefix);
    if (match) atomicAdd(&h[(u32)(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  if (h[threadIdx.x]) atomicAdd(&hist256[threadIdx.x], h[threadIdx.x]);
}

// the bucket that holds the need-th largest key of this digit; leaves the histogram zeroed for the next pass
__global__ __launch_bounds__(256) void k_select_step(SelectState* st, u32* hist256) {
  __shared__ u32 h[256];
  h[threadIdx.x] = hist256[threadIdx.x];
  hist256[threadIdx.x] = 0;
  __syncthreads();
  if (threadIdx.x != 0) return;
  SelectState s = *st;
  i64 cum = 0;
  int b = 255;
  for (; b > 0; b--) {
    if (cum + h[b] >= s.need) break;
    cum += h[b];
  }
  s.greater += cum;
  s.need -= cum;
  s.eq_count = h[b];
  s.prefix = (s.prefix << 8) | (u64)b;
  *st = s;
}

hipError_t launch_radix_select(const uint64_t* key, int64_t count, int64_t need, uint32_t* hist256, SelectState* st,
                               hipStream_t stream) {
  const int grid = (int)hmin((count + 2047) / 2048, 2048);
  hipLaunchKernelGGL(k_select_init, dim3(1), dim3(256), 0, stream, st, need, hist256);
  for (int shift = 56; shift >= 0; shift -= 8) {
    hipLaunchKernelGGL(k_hist_st, dim3(grid > 0 ? grid : 1), dim3(256), 0, stream, key, count, shift, st, hist256);
    hipLaunchKernelGGL(k_select_step, dim3(1), dim3(256), 0, stream, st, hist256);
  }
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_collect_gt(const u64* key, i64 count, u64 thr, u32* out, u32* n_out, u32 cap) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (i64)gridDim.x * blockDim.x) {
    if (key[i] > thr) {
      const u32 pos = atomicAdd(n_out, 1u);
      if (pos < cap) out[pos] = (u32)i;
    }
  }
}

hipError_t launch_collect_gt(const uint64_t* key, int64_t count, uint64_t thr, uint32_t* out, uint32_t* n_out,
                             uint32_t cap, hipStream_t stream) {
  const int grid = (int)hmin((count + 2047) / 2048, 2048);
  hipLaunchKernelGGL(k_collect_gt, dim3(grid > 0 ? grid : 1), dim3(256), 0, stream, key, count, thr, out, n_out, cap);
  return hipGetLastError();
}

constexpr int kEqChunk = 1024;   // entries per wave

__global__ __launch_bounds__(256) void k_eq_count(const u64* key, i64 count, u64 thr, u32* chunk_cnt) {
  const int lane = threadIdx.x & 63;
  const i64 chunk = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i64 base = chunk * kEqChunk;
  if (base >= count) return;
  u32 c = 0;
  for (int o = lane; o < kEqChunk; o += 64) {
    const i64 i = base + o;
    if (i < count && key[i] == thr) c++;
  }
  c = wave_sum(c);
  if (lane == 0) chunk_cnt[chunk] = c;
}

hipError_t launch_eq_count(const uint64_t* key, int64_t count, uint64_t thr, uint32_t* chunk_cnt, hipStream_t stream) {
  const i64 chunks = (count + kEqChunk - 1) / kEqChunk;
  if (chunks == 0) return hipSuccess;
  hipLaunchKernelGGL(k_eq_count, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, stream, key, count, thr, chunk_cnt);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_eq_collect(const u64* key, i64 count, u64 thr, const u32* chunk_base, u32 m,
                                                    u32* out) {
  const int lane = threadIdx.x & 63;
  const i64 chunk = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i64 base = chunk * kEqChunk;
  if (base >= count) return;
  u32 rank = chunk_base[chunk];
  if (rank >= m) return;
  for (int o = 0; o < kEqChunk; o += 64) {
    const i64 i = base + o + lane;
    const bool eq = (i < count) && (key[i] == thr);
    const u64 ball = __ballot(eq);
    const u32 before = __popcll(ball & (((u64)1 << lane) - 1));
    if (eq && rank + before < m) out[rank + before] = (u32)i;
    rank += __popcll(ball);
    if (rank >= m) return;
  }
}

hipError_t launch_eq_collect(const uint64_t* key, int64_t count, uint64_t thr, const uint32_t* chunk_base, uint32_t m,
                             uint32_t* out, hipStream_t stream) {
  const i64 chunks = (count + kEqChunk - 1) / kEqChunk;
  if (chunks == 0 || m == 0) return hipSuccess;
  hipLaunchKernelGGL(k_eq_collect, dim3((unsigned)((chunks + 3) / 4)), dim3(256), 0, stream, key, count, thr,
                     chunk_base, m, out);
  return hipGetLastError();
}

__global__ void k_gather_winners(const u32* sel, u32 nsel, const u64* key, const u32* cases, const u32* ctrls,
                                 const u32* row0, const u32* row1, u64* o_key, u32* o_cases, u32* o_ctrls, u32* o_row0,
                                 u32* o_row1) {
  const u32 i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nsel) return;
  const u32 p = sel[i];
  o_key[i] = key[p];
  o_cases[i] = cases[p];
  o_ctrls[i] = ctrls[p];
  o_row0[i] = row0[p];
  o_row1[i] = row1[p] & 0x7fffffffu;
}

hipError_t launch_gather_winners(const uint32_t* sel, uint32_t nsel, const uint64_t* key, const uint32_t* cases,
                                 const uint32_t* ctrls, const uint32_t* row0, const uint32_t* row1, uint64_t* o_key,
                                 uint32_t* o_cases, uint32_t* o_ctrls, uint32_t* o_row0, uint32_t* o_row1,
                                 hipStream_t stream) {
  if (nsel == 0) return hipSuccess;
  hipLaunchKernelGGL(k_gather_winners, dim3((nsel + 255) / 256), dim3(256), 0, stream, sel, nsel, key, cases, ctrls,
                     row0, row1, o_key, o_cases, o_ctrls, o_row0, o_row1);
  return hipGetLastError();
}

__global__ void k_fill_u32(u32* p, i64 n, u32 v) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) p[i] = v;
}

hipError_t launch_fill_u32(uint32_t* p, int64_t n, uint32_t v, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  const int grid = (int)hmin((n + 255) / 256, 4096);
  hipLaunchKernelGGL(k_fill_u32, dim3(grid), dim3(256), 0, stream, p, n, v);
  return hipGetLastError();
}

__global__ void k_max_merge(u32* dst, const u32* src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = max(dst[i], src[i]);
}

hipError_t launch_max_merge(uint32_t* dst, const uint32_t* src, int n, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_max_merge, dim3((n + 255) / 256), dim3(256), 0, stream, dst, src, n);
  return hipGetLastError();
}

}  // namespace gcre
