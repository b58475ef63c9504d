// gcre_ie.hip -- the inclusion-exclusion form of the permutation (null) kernel for gfx950, and the count planes
// it runs on.
//
// Same result as k_null / k_null_sparse -- count[p][r] = popc(joined_p & mask_r), reference src/methods.h:73-88 --
// but a joined path is never walked bit by bit.  Every path set keeps, next to its rows, its COUNT PLANES
//
//     N[row][r] = popc(row & mask_r)        bit-sliced: plane p of tile kt holds bit p of 2048 counts
//
// and a joined path  p0[idx] | z  (z = the row the join adds: one gene at levels 1-4, a 2-gene path at level 5) is
//
//     count = N0[idx] + Nz[z] - popc(p0[idx] & z & mask_r)
//
// N0[idx] is loaded once per uid (all `count` joins of a uid share paths0[idx], join_base.cpp:242), Nz[z] is a
// couple of wide loads, and only the OVERLAP p0[idx] & z -- a handful of patients for rare variants -- is still
// streamed as transposed mask rows through the carry-save tree of gcre_bitslice.h.  Paths whose overlap is longer
// than what the join adds fall back, per path, to streaming the delta z & ~p0 (what k_null_sparse always does).
// A kept join writes the planes of its joined paths as a by-product: they are the N0 of the next level.
//
// The epilogue is pruned exactly: the running maxima only move when T[total][count] exceeds them, and for a
// threshold theta no larger than any running maximum of the tile the counts that can matter lie outside an interval
// [lo, hi] of the path's table diagonal (precomputed ladder, k_build_ladder).  Two bit-sliced borrow chains test
// all 2048 counts of a path against (lo, hi); only when some permutation falls outside does the wave transpose its
// counters and look the table up (finish_m1).  Skipping a lookup whose value cannot exceed the maximum leaves every
// maximum bit-identical.
#include "gcre_bitslice.h"
#include "gcre_kernels.h"

namespace gcre {

constexpr int kIeWaves = 4;
constexpr int kIeDiagCap = 1024;
constexpr int kIeDiagCap2 = 512;
constexpr int kIeRefresh = 32;      // segments between two reads of the global maxima (threshold refresh)

__device__ __forceinline__ u32 maj3(u32 a, u32 b, u32 c) { return (a & b) | ((a ^ b) & c); }

__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u32 t = (u32)__shfl_xor((int)v, o, 64);
    v = t < v ? t : v;
  }
  return v;
}

// L = counter planes of the joined paths, a multiple of 4.
template <int M, int L>
__global__ __launch_bounds__(64 * kIeWaves) void k_null_ie(const IeArgs a) {
  static_assert(L % 4 == 0 && L >= 8 && L <= 16, "planes come in groups of 4");
  __shared__ __attribute__((aligned(8))) u32 diag_lds[kIeWaves][kIeDiagCap];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const i64 wi = (i64)(blockIdx.x >> 3) * kIeWaves + wave;
  const i64 wx = a.waves_per_xcd;
  const i64 slices = 8 * wx;
  const u32 lane4 = (u32)lane * 4u;
  u32* dl = diag_lds[wave];

  const SparseSeg GCRE_CONSTANT* segs = (const SparseSeg GCRE_CONSTANT*)a.segs;
  const u64 GCRE_CONSTANT* loff0 = (const u64 GCRE_CONSTANT*)a.loff0;
  const u32 GCRE_CONSTANT* lidx0 = (const u32 GCRE_CONSTANT*)a.lidx0;
  const u64 GCRE_CONSTANT* doff = (const u64 GCRE_CONSTANT*)a.doff;
  const u32 GCRE_CONSTANT* dlist = (const u32 GCRE_CONSTANT*)a.dlist;
  const u32 GCRE_CONSTANT* tots = (const u32 GCRE_CONSTANT*)a.tot;
  const u32 GCRE_CONSTANT* rowz = (const u32 GCRE_CONSTANT*)a.rowz;
  const u32 GCRE_CONSTANT* ladder = (const u32 GCRE_CONSTANT*)a.ladder;

  u32 nmax[32];
#pragma unroll
  for (int q = 0; q < 32; q++) nmax[q] = 0u;
  int cur_kt = -1;
  u32 valid = 0u;       // bit q: permutation 32*lane + q of the tile exists (< K)
  u32 theta = 0u;       // wave-uniform: no running maximum of the tile is below this (f32 bit pattern)
  u32 lad_base = 0u;    // ladder row of the threshold level in use
  __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)a.mt, 0, 0x7fffffff, 0x00020000);

  // push the wave's maxima to the global array, pull what the other waves found, and set the pruning threshold to
  // the smallest running maximum of the tile's live permutations (stale reads only make it smaller: still exact)
  auto exchange = [&]() {
    u32* out = a.null_bits + (size_t)cur_kt * 2048 + lane * 32;
    u32 lo = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 32; q++) {
      const u32 g = __hip_atomic_load(out + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // past the L1
      if (nmax[q] > g) atomicMax(out + q, nmax[q]);
      const u32 v = nmax[q] > g ? nmax[q] : g;
      if ((valid >> q) & 1u) lo = v < lo ? v : lo;
    }
    theta = __builtin_amdgcn_readfirstlane(wave_min_u32(lo));
    if (theta == 0xffffffffu) theta = 0u;
    // level j covers thresholds >= j / kLadderPerUnit
    const float tf = __uint_as_float(theta);
    int j = (int)(tf * (float)kLadderPerUnit);
    j = j < 0 ? 0 : (j > kLadderLevels - 1 ? kLadderLevels - 1 : j);
    lad_base = (u32)j * (u32)a.ladder_stride;
  };
  auto flush = [&]() {
    if (cur_kt < 0) return;
    u32* out = a.null_bits + (size_t)cur_kt * 2048 + lane * 32;
#pragma unroll
    for (int q = 0; q < 32; q++) {
      if (nmax[q] != 0u) atomicMax(out + q, nmax[q]);
      nmax[q] = 0u;
    }
  };

  auto to_counts = [&](const u32 (&C)[L], u32 (&R)[16]) {
#pragma unroll
    for (int l = 0; l < 16; l++) R[l] = (l < L) ? C[l] : 0u;
    transpose16(R);
  };
  auto wave_lds_fence = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  // method 1: counts -> f32 table diagonal -> running maxima (methods.h:96-103)
  auto finish_m1 = [&](const u32 (&C)[L], u32 total) {
    u32 R[16];
    to_counts(C, R);
    const u32* diag_g = (const u32*)a.t32 + sp_diag_offset(total);
    if (total < (u32)kIeDiagCap) {
      for (u32 i = (u32)lane; i <= total; i += 64) dl[i] = diag_g[i];
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const u32 lo = dl[R[j] & 0xffffu];
        const u32 hi = dl[R[j] >> 16];
        nmax[j] = (lo > nmax[j]) ? lo : nmax[j];
        nmax[j + 16] = (hi > nmax[j + 16]) ? hi : nmax[j + 16];
      }
      __builtin_amdgcn_wave_barrier();
    } else {
      for (int j = 0; j < 16; j++) {
        const u32 lo = diag_g[R[j] & 0xffffu];
        const u32 hi = diag_g[R[j] >> 16];
        nmax[j] = (lo > nmax[j]) ? lo : nmax[j];
        nmax[j + 16] = (hi > nmax[j + 16]) ? hi : nmax[j + 16];
      }
    }
  };

  // method 2: vtmax[a][tp-a] + vtmax[tn-b][b] in f64, rounded to f32, clamped at 0 (methods.h:220-230)
  auto finish_m2 = [&](const u32 (&Cp)[L], const u32 (&Cn)[L], u32 tp, u32 tn) {
    u32 Rp[16], Rn[16];
    to_counts(Cp, Rp);
    to_counts(Cn, Rn);
    const double* dp = a.d64 + sp_diag_offset(tp);
    const double* dn = a.d64 + sp_diag_offset(tn);
    const bool staged = (tp < (u32)kIeDiagCap2 / 2) && (tn < (u32)kIeDiagCap2 / 2);
    double* lp = (double*)dl;
    double* ln = lp + kIeDiagCap2 / 2;
    if (staged) {
      for (u32 i = (u32)lane; i <= tp; i += 64) lp[i] = dp[i];
      for (u32 i = (u32)lane; i <= tn; i += 64) ln[i] = dn[i];
      wave_lds_fence();
    }
    auto one = [&](u32 ca, u32 cb, u32& m) {
      const double s = staged ? (lp[ca] + ln[cb]) : (dp[ca] + dn[cb]);
      float f = (float)s;
      f = (f > 0.0f) ? f : 0.0f;
      const u32 v = __float_as_uint(f);
      m = (v > m) ? v : m;
    };
#pragma unroll
    for (int j = 0; j < 16; j++) {
      one(Rp[j] & 0xffffu, Rn[j] & 0xffffu, nmax[j]);
      one(Rp[j] >> 16, Rn[j] >> 16, nmax[j + 16]);
    }
    if (staged) __builtin_amdgcn_wave_barrier();
  };

  // some live permutation has a count outside [lo, hi]?  Two borrow chains over the planes, scalar bound bits:
  // C < lo  <=>  C - lo borrows;  C > hi  <=>  hi - C borrows.
  auto outside = [&](const u32 (&C)[L], u32 lo, u32 hi) -> bool {
    u32 blo = 0u, bhi = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
      const u32 kl = (u32)(-(int)((lo >> l) & 1u));   // scalar: all ones when the bound has bit l
      const u32 kh = (u32)(-(int)((hi >> l) & 1u));
      blo = maj3(~C[l], kl, blo);
      bhi = maj3(~kh, C[l], bhi);
    }
    return __builtin_amdgcn_ballot_w64(((blo | bhi) & valid) != 0u) != 0ull;
  };

  // count planes of one (row-half, tile): groups of 4 planes, [group][lane][4] dwords
  auto load_planes = [&](u32 (&P)[L], const u32* planes, u64 rowhalf, int groups) {
    const u32x4* src = (const u32x4*)(planes + ((rowhalf * (u64)a.nkt + (u64)cur_kt) * (u64)groups) * 256u) + lane;
#pragma unroll
    for (int j = 0; j < L / 4; j++) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (j < groups) v = src[j * 64];
      P[4 * j + 0] = v.x;
      P[4 * j + 1] = v.y;
      P[4 * j + 2] = v.z;
      P[4 * j + 3] = v.w;
    }
  };

  for (int step = 0; step < a.nkt; step++) {
    const i64 item = (i64)xcd * a.nkt * wx + wi + (i64)step * wx;
    const int kt = (int)(item / slices);
    const i64 sl = item % slices;
    if (kt != cur_kt) {
      flush();
      cur_kt = kt;
      mt = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mt + (size_t)kt * a.mt_rows * 64), 0, 0x7fffffff, 0x00020000);
      const int live = a.K - kt * 2048 - lane * 32;           // permutations of this lane that exist
      valid = live >= 32 ? 0xffffffffu : (live <= 0 ? 0u : ((1u << live) - 1u));
      theta = 0u;
      lad_base = 0u;
    }
    int since = kIeRefresh;   // refresh right away: pick up what earlier waves already published
    for (i64 sidx = sl; sidx < a.nsegs; sidx += slices) {
      const u32 row0 = segs[sidx].row0;
      const u32 first = segs[sidx].first;
      const u32 npaths = segs[sidx].n;
      if (a.prune && ++since > kIeRefresh) {
        exchange();
        since = 0;
      }

      u32 x[16];
      auto stream = [&](u32 (&P)[L], const u32 GCRE_CONSTANT* list, u64 p, u64 e) {
        for (; p + 16 <= e; p += 16) {
          load16(x, mt, lane4, *(const u32x16 GCRE_CONSTANT*)(list + p));
          add16<L>(P, x);
        }
        for (; p < e; p += 4) {
          const u32x4 offs = *(const u32x4 GCRE_CONSTANT*)(list + p);
          u32 y[4];
#pragma unroll
          for (int j = 0; j < 4; j++) y[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
          add4<L>(P, y);
        }
      };

      // ---- base counters: the planes of paths0[row0], or its bits streamed when the set has no planes ----
      u32 B[M][L];
#pragma unroll
      for (int h = 0; h < M; h++) {
        const u64 r = (u64)row0 * M + h;
        if (a.planes0) {
          load_planes(B[h], a.planes0, r, a.g0);
        } else {
#pragma unroll
          for (int l = 0; l < L; l++) B[h][l] = 0u;
          stream(B[h], lidx0, loff0[r], loff0[r + 1]);
        }
      }

      for (u32 t = 0; t < npaths; t++) {
        const u32 q = first + t;
        const u32 rz = rowz[q];
        u32 C[M][L];
#pragma unroll
        for (int h = 0; h < M; h++) {
          const u64 d = (u64)q * M + h;
          const u64 o0 = doff[d], o1 = doff[d + 1];
          const u64 lb = o0 & ~(u64)3, le = o1 & ~(u64)3;
          if ((o0 & 1u) == 0u) {
            // delta list: the bits the join adds on top of paths0
#pragma unroll
            for (int l = 0; l < L; l++) C[h][l] = B[h][l];
            stream(C[h], dlist, lb, le);
          } else {
            // overlap list: C = B + Nz - popc(p0 & z & mask)
            u32 S[L], Z[L];
#pragma unroll
            for (int l = 0; l < L; l++) S[l] = 0u;
            const int hz = (M == 2 && (rz >> 31)) ? 1 - h : h;
            load_planes(Z, a.planesz, (u64)(rz & 0x7fffffffu) * M + hz, a.gz);
            stream(S, dlist, lb, le);
            u32 cy = 0u, bw = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 s1 = B[h][l] ^ Z[l] ^ cy;
              cy = maj3(B[h][l], Z[l], cy);
              C[h][l] = s1 ^ S[l] ^ bw;
              bw = maj3(~s1, S[l], bw);
            }
          }
        }
        if (a.planes_out) {
#pragma unroll
          for (int h = 0; h < M; h++) {
            const u64 rh = ((u64)a.out_first + q) * M + h;
            u32x4* dst = (u32x4*)(a.planes_out + ((rh * (u64)a.nkt + (u64)cur_kt) * (u64)a.go) * 256u) + lane;
#pragma unroll
            for (int j = 0; j < 4; j++) {
              if (j < a.go) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (4 * j < L) v = u32x4{C[h][(4 * j) % L], C[h][(4 * j + 1) % L], C[h][(4 * j + 2) % L], C[h][(4 * j + 3) % L]};
                dst[j * 64] = v;
              }
            }
          }
        }
        if (q < a.score_begin || q >= a.score_end) continue;   // planes only: the path belongs to another shard
        if constexpr (M == 1) {
          const u32 total = tots[q];
          if (a.prune) {
            const u32 lh = ladder[lad_base + total];
            if (!outside(C[0], lh & 0xffffu, lh >> 16)) continue;
          }
          finish_m1(C[0], total);
        } else {
          finish_m2(C[0], C[M - 1], tots[2 * q], tots[2 * q + 1]);
        }
      }
    }
  }
  flush();
}

#define GCRE_IE_DISPATCH(EXPR)                             \
  if (method == 1) {                                       \
    if (planes <= 8) { EXPR(1, 8); }                       \
    else if (planes <= 12) { EXPR(1, 12); }                \
    else { EXPR(1, 16); }                                  \
  } else {                                                 \
    if (planes <= 8) { EXPR(2, 8); }                       \
    else if (planes <= 12) { EXPR(2, 12); }                \
    else { EXPR(2, 16); }                                  \
  }

hipError_t launch_null_ie(const IeArgs& a, int method, int planes, hipStream_t stream) {
  const dim3 grid((unsigned)(8 * a.waves_per_xcd / kIeWaves));
  const dim3 block(64 * kIeWaves);
#define GCRE_LAUNCH(MM, LL) hipLaunchKernelGGL((k_null_ie<MM, LL>), grid, block, 0, stream, a)
  GCRE_IE_DISPATCH(GCRE_LAUNCH)
#undef GCRE_LAUNCH
  return hipGetLastError();
}

int ie_max_waves_per_cu(int method, int planes) {
  int blocks = 0;
  hipError_t e = hipSuccess;
#define GCRE_OCC(MM, LL) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_null_ie<MM, LL>, 64 * kIeWaves, 0)
  GCRE_IE_DISPATCH(GCRE_OCC)
#undef GCRE_OCC
  if (e != hipSuccess || blocks < 1) blocks = 1;
  return blocks * kIeWaves;
}

// ------------------------------------------------------------------------------------------------
// count planes of a path set from its bit lists: one wave per (row-half, tile)
// ------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256) void k_build_planes(const u32* mt_all, u32 mt_rows, int nkt, const u64* loff,
                                                      const u32* lidx, i64 nrowhalves, int groups, u32* planes) {
  const int lane = threadIdx.x & 63;
  const i64 wave = (i64)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform
  const i64 nwaves = (i64)gridDim.x * 4;
  const u32 lane4 = (u32)lane * 4u;
  const u64 GCRE_CONSTANT* off = (const u64 GCRE_CONSTANT*)loff;
  const u32 GCRE_CONSTANT* list = (const u32 GCRE_CONSTANT*)lidx;
  const i64 items = nrowhalves * nkt;
  // tile-major so that concurrently running waves share a mask tile in L2
  for (i64 it = wave; it < items; it += nwaves) {
    const i64 w = it / nrowhalves;   // tile
    const i64 rr = it - w * nrowhalves;
    __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)(mt_all + (size_t)w * mt_rows * 64), 0, 0x7fffffff, 0x00020000);
    u32 P[L];
#pragma unroll
    for (int l = 0; l < L; l++) P[l] = 0u;
    u64 p = off[rr];
    const u64 e = off[rr + 1];
    u32 x[16];
    for (; p + 16 <= e; p += 16) {
      load16(x, mt, lane4, *(const u32x16 GCRE_CONSTANT*)(list + p));
      add16<L>(P, x);
    }
    for (; p < e; p += 4) {
      const u32x4 offs = *(const u32x4 GCRE_CONSTANT*)(list + p);
      u32 y[4];
#pragma unroll
      for (int j = 0; j < 4; j++) y[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
      add4<L>(P, y);
    }
    u32x4* dst = (u32x4*)(planes + (((u64)rr * (u64)nkt + (u64)w) * (u64)groups) * 256u) + lane;
#pragma unroll
    for (int j = 0; j < L / 4; j++)
      if (j < groups) dst[j * 64] = u32x4{P[4 * j], P[4 * j + 1], P[4 * j + 2], P[4 * j + 3]};
  }
}

hipError_t launch_build_planes(const uint32_t* mt, uint32_t mt_rows, int nkt, const uint64_t* loff, const uint32_t* lidx,
                               int64_t nrowhalves, int groups, uint32_t* planes, hipStream_t stream) {
  const i64 items = nrowhalves * nkt;
  if (items == 0) return hipSuccess;
  const i64 blocks = (items + 3) / 4;
  const dim3 grid((unsigned)(blocks < 256 * 8 ? blocks : 256 * 8)), block(256);
  if (groups <= 2) hipLaunchKernelGGL(k_build_planes<8>, grid, block, 0, stream, mt, mt_rows, nkt, loff, lidx, nrowhalves, groups, planes);
  else if (groups == 3) hipLaunchKernelGGL(k_build_planes<12>, grid, block, 0, stream, mt, mt_rows, nkt, loff, lidx, nrowhalves, groups, planes);
  else hipLaunchKernelGGL(k_build_planes<16>, grid, block, 0, stream, mt, mt_rows, nkt, loff, lidx, nrowhalves, groups, planes);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pruning ladder of the method-1 null table: for every diagonal t and threshold level j the largest interval
// [lo, hi] around the diagonal's minimum on which T32[t][c] <= j / kLadderPerUnit.  Entry = hi << 16 | lo;
// an empty interval is lo = 1, hi = 0 (every count is "outside").  Valid for ANY table: cells outside the interval
// are simply looked up.
// ------------------------------------------------------------------------------------------------
__global__ void k_build_ladder(const u32* t32, int TD, u32* ladder) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= TD) return;
  const u32* d = t32 + sp_diag_offset((u32)t);
  int c0 = 0;
  u32 best = d[0];
  for (int c = 1; c <= t; c++)
    if (d[c] < best) { best = d[c]; c0 = c; }
  int lo = c0 + 1, hi = c0;   // empty
  for (int j = 0; j < kLadderLevels; j++) {
    const u32 th = __float_as_uint((float)j / (float)kLadderPerUnit);
    if (lo > hi && best <= th) lo = hi = c0;
    if (lo <= hi) {
      while (lo > 0 && d[lo - 1] <= th) lo--;
      while (hi < t && d[hi + 1] <= th) hi++;
      ladder[(size_t)j * TD + t] = ((u32)hi << 16) | (u32)lo;
    } else {
      ladder[(size_t)j * TD + t] = 1u;   // lo = 1, hi = 0
    }
  }
}

hipError_t launch_build_ladder(const float* t32, int TD, uint32_t* ladder, hipStream_t stream) {
  hipLaunchKernelGGL(k_build_ladder, dim3((unsigned)((TD + 63) / 64)), dim3(64), 0, stream, (const u32*)t32, TD, ladder);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// inspector: per joined path (and half) the list the kernel streams -- the bits of the reduced row z that are
// clear in the paths0 row (mode 0, delta) or set in it (mode 1, overlap); the mode sits in bit 0 of doff
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ie_fill(const u32* p0, int S32, int W32p, int M, const u32* row0, const u32* rowz,
                                                 i64 count, const u64* loffz, const u32* lidxz, const u64* doff,
                                                 u32 zoff, u32* dlist) {
  const int lane = threadIdx.x & 63;
  const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
  for (i64 w = wave; w < count * M; w += nwaves) {
    const i64 i = w / M;
    const int h = (int)(w % M);
    const u32* r0 = p0 + (size_t)row0[i] * S32 + (size_t)h * W32p;
    const u32 rzraw = rowz[i];
    const u32 rz = rzraw & 0x7fffffffu;
    const int hz = (M == 2 && (rzraw >> 31)) ? 1 - h : h;
    const u64 li = (u64)rz * M + hz;
    const u64 b1 = loffz[li], e1 = loffz[li + 1];
    const u64 o0 = doff[w];
    const bool want_set = (o0 & 1u) != 0u;
    u64 out = o0 & ~(u64)3;
    const u64 out_end = doff[w + 1] & ~(u64)3;
    for (u64 p = b1; p < e1; p += 64) {
      const u32 e = (p + lane < e1) ? lidxz[p + lane] : zoff;
      const u32 idx = e >> 8;
      bool keep = (e != zoff);
      const u32 w0 = keep ? r0[idx >> 5] : 0u;
      keep = keep && ((((w0 >> (idx & 31u)) & 1u) != 0u) == want_set);
      const u64 m = __ballot(keep);
      const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
      if (keep) dlist[out + before] = e;
      out += (u64)__builtin_popcountll(m);
    }
    if (out + lane < out_end) dlist[out + lane] = zoff;
  }
}

hipError_t launch_ie_fill(const uint32_t* p0, int S32, int W32p, int method, const uint32_t* row0, const uint32_t* rowz,
                          int64_t count, const uint64_t* loffz, const uint32_t* lidxz, const uint64_t* doff, uint32_t zoff,
                          uint32_t* dlist, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const i64 blocks = (count * method + 3) / 4;
  hipLaunchKernelGGL(k_ie_fill, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, p0, S32, W32p,
                     method, row0, rowz, count, loffz, lidxz, doff, zoff, dlist);
  return hipGetLastError();
}

}  // namespace gcre
