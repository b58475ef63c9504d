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
// Set bits come from per-row index lists kept next to every path set (CSR: offsets + pre-scaled row offsets
// into the mask tile, padded to blocks of 16 with the all-zero row).  They are wave-uniform, so they travel on
// the scalar side: s_load_dwordx16 fetches 16 offsets, each becomes the scalar offset of one
// buffer_load_dword (lane*4 is the vector offset) -- no VALU, no LDS for addressing.
//
// Joins that share their paths0 row (all `count` joins of one uid, join_base.cpp:242) share its bits: the
// wave accumulates the bits of paths0[idx] once into base counters and per joined path only adds the bits
// that paths1[loc] contributes on top: the entries of paths1's list whose bit is clear in paths0's row,
// compacted with a ballot and pulled lane by lane (v_readlane) onto the scalar side.
//
// Per joined path the counters are transposed in-register (16x16 bit-matrix transpose on both halves of
// every dword) into 32 integers, looked up on the path's table diagonal and max-ed into 32 running maxima
// per lane.  Waves are independent: no barriers, no LDS.
#include "gcre_bitslice.h"
#include "gcre_kernels.h"

namespace gcre {

constexpr int kSparseWaves = 4;           // waves per block, each fully independent
constexpr int kDiagCap = 1024;   // table-diagonal entries staged in LDS per wave (longer diagonals are gathered from L2)
constexpr int kDiagCap2 = 512;   // method 2: f64 cells staged per half

// M = 1: one counter set per joined path (unsigned method, methods.h:58-105).
// M = 2: the (+) and (-) halves are two independent sparse accumulations (methods.h:130-232); half h of joined
//        path q uses list 2*q+h, half h of paths0 row r uses list 2*r+h.
template <int M, int L>
__global__ __launch_bounds__(64 * kSparseWaves) void k_null_sparse(const SparseArgs a) {
  __shared__ __attribute__((aligned(8))) u32 diag_lds[kSparseWaves][kDiagCap];   // M=2: 2 x 256 doubles
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
  const u32 lane4 = (u32)lane * 4u;
  u32* dl = diag_lds[wave];

  const SparseSeg GCRE_CONSTANT* segs = (const SparseSeg GCRE_CONSTANT*)a.segs;
  const u64 GCRE_CONSTANT* loff0 = (const u64 GCRE_CONSTANT*)a.loff0;
  const u32 GCRE_CONSTANT* lidx0 = (const u32 GCRE_CONSTANT*)a.lidx0;
  const u64 GCRE_CONSTANT* doff = (const u64 GCRE_CONSTANT*)a.doff;
  const u32 GCRE_CONSTANT* dlist = (const u32 GCRE_CONSTANT*)a.dlist;
  const u32 GCRE_CONSTANT* tots = (const u32 GCRE_CONSTANT*)a.tot;

  u32 nmax[32];
#pragma unroll
  for (int q = 0; q < 32; q++) nmax[q] = 0u;
  int cur_kt = -1;
  __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)a.mt, 0, 0x7fffffff, 0x00020000);

  auto flush = [&]() {
    if (cur_kt < 0) return;
    u32* out = a.null_bits + (size_t)cur_kt * 2048 + lane * 32;
#pragma unroll
    for (int q = 0; q < 32; q++) {
      if (nmax[q] != 0u) atomicMax(out + q, nmax[q]);
      nmax[q] = 0u;
    }
  };

  // counter planes -> 16 registers of packed 16-bit counts: permutation j in the low half, j+16 in the high half
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
    if (total < (u32)kDiagCap) {
      // the diagonal (total+1 cells) goes through LDS: a few coalesced loads instead of 32 64-address gathers
      for (u32 i = (u32)lane; i <= total; i += 64) dl[i] = diag_g[i];
      wave_lds_fence();
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const u32 lo = dl[R[j] & 0xffffu];
        const u32 hi = dl[R[j] >> 16];
        nmax[j] = (lo > nmax[j]) ? lo : nmax[j];
        nmax[j + 16] = (hi > nmax[j + 16]) ? hi : nmax[j + 16];
      }
      __builtin_amdgcn_wave_barrier();   // the next path overwrites dl
    } else {
      for (int j = 0; j < 16; j++) {
        const u32 lo = diag_g[R[j] & 0xffffu];
        const u32 hi = diag_g[R[j] >> 16];
        nmax[j] = (lo > nmax[j]) ? lo : nmax[j];
        nmax[j + 16] = (hi > nmax[j + 16]) ? hi : nmax[j + 16];
      }
    }
  };

  // method 2: vtmax[a][tp-a] + vtmax[tn-b][b] in f64, rounded to f32, clamped at the 0 the maxima start from
  // (methods.h:220-230); a, b = counts of the (+) / (-) half
  auto finish_m2 = [&](const u32 (&Cp)[L], const u32 (&Cn)[L], u32 tp, u32 tn) {
    u32 Rp[16], Rn[16];
    to_counts(Cp, Rp);
    to_counts(Cn, Rn);
    const double* dp = a.d64 + sp_diag_offset(tp);
    const double* dn = a.d64 + sp_diag_offset(tn);
    const bool staged = (tp < (u32)kDiagCap2 / 2) && (tn < (u32)kDiagCap2 / 2);
    double* lp = (double*)dl;
    double* ln = lp + kDiagCap2 / 2;
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

  for (int step = 0; step < a.nkt; step++) {
    const i64 item = (i64)xcd * a.nkt * wx + wi + (i64)step * wx;
    const int kt = (int)(item / slices);
    const i64 sl = item % slices;
    if (kt != cur_kt) {
      flush();
      cur_kt = kt;
      mt = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mt + (size_t)kt * a.mt_rows * 64), 0, 0x7fffffff, 0x00020000);
    }
    for (i64 sidx = sl; sidx < a.nsegs; sidx += slices) {
      const u32 row0 = segs[sidx].row0;
      const u32 first = segs[sidx].first;
      const u32 npaths = segs[sidx].n;

      // Every phase is a stream of 16-entry blocks of wave-uniform row offsets (s_load_dwordx16 -> 16 scalar
      // offsets -> 16 buffer loads -> adder tree).
      u32 x[16];
      auto stream = [&](u32 (&P)[L], const u32 GCRE_CONSTANT* list, u64 p, u64 e) {
        for (; p + 16 <= e; p += 16) {
          load16(x, mt, lane4, *(const u32x16 GCRE_CONSTANT*)(list + p));
          add16<L>(P, x);
        }
        for (; p < e; p += 4) {          // tail: blocks of 4 (the mask-row loads are what bounds this kernel)
          const u32x4 offs = *(const u32x4 GCRE_CONSTANT*)(list + p);
          u32 y[4];
#pragma unroll
          for (int j = 0; j < 4; j++) y[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
          add4<L>(P, y);
        }
      };

      // ---- bits of the shared paths0 row (each half for method 2) -> base counters ----
      u32 B[M][L];
#pragma unroll
      for (int h = 0; h < M; h++) {
#pragma unroll
        for (int l = 0; l < L; l++) B[h][l] = 0u;
        const u64 r = (u64)row0 * M + h;
        stream(B[h], lidx0, loff0[r], loff0[r + 1]);
      }

      // ---- per joined path: the bits paths1 adds on top of paths0 (delta lists, built once per join) ----
      for (u32 t = 0; t < npaths; t++) {
        const u32 q = first + t;
        u32 C[M][L];
#pragma unroll
        for (int h = 0; h < M; h++) {
#pragma unroll
          for (int l = 0; l < L; l++) C[h][l] = B[h][l];
          const u64 d = (u64)q * M + h;
          stream(C[h], dlist, doff[d], doff[d + 1]);
        }
        if constexpr (M == 1) finish_m1(C[0], tots[q]);
        else finish_m2(C[0], C[M - 1], tots[2 * q], tots[2 * q + 1]);
      }
    }
  }
  flush();
}

#define GCRE_SPARSE_DISPATCH(EXPR)                         \
  if (method == 1) {                                       \
    if (planes <= 8) { EXPR(1, 8); }                       \
    else if (planes <= 10) { EXPR(1, 10); }                \
    else if (planes <= 12) { EXPR(1, 12); }                \
    else { EXPR(1, 16); }                                  \
  } else {                                                 \
    if (planes <= 8) { EXPR(2, 8); }                       \
    else if (planes <= 10) { EXPR(2, 10); }                \
    else if (planes <= 12) { EXPR(2, 12); }                \
    else { EXPR(2, 16); }                                  \
  }

hipError_t launch_null_sparse(const SparseArgs& a, int method, int planes, hipStream_t stream) {
  const dim3 grid((unsigned)(8 * a.waves_per_xcd / kSparseWaves));
  const dim3 block(64 * kSparseWaves);
#define GCRE_LAUNCH(MM, LL) hipLaunchKernelGGL((k_null_sparse<MM, LL>), grid, block, 0, stream, a)
  GCRE_SPARSE_DISPATCH(GCRE_LAUNCH)
#undef GCRE_LAUNCH
  return hipGetLastError();
}

int sparse_max_waves_per_cu(int method, int planes) {
  int blocks = 0;
  hipError_t e = hipSuccess;
#define GCRE_OCC(MM, LL) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_null_sparse<MM, LL>, 64 * kSparseWaves, 0)
  GCRE_SPARSE_DISPATCH(GCRE_OCC)
#undef GCRE_OCC
  if (e != hipSuccess || blocks < 1) blocks = 1;
  return blocks * kSparseWaves;
}

// ------------------------------------------------------------------------------------------------
// inspector: per joined path the list of bits that paths1 adds on top of paths0 (once per join, not per tile)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_delta_fill(const u32* p0, int S32, int W32p, int M, const u32* row0,
                                                    const u32* row1, i64 count, const u64* loff1, const u32* lidx1,
                                                    const u64* doff, u32 zoff, u32* dlist) {
  // one wave per (joined path, half): half h of the joined path takes half h (or 1-h when the relation flips the
  // sign, methods.h:140-142) of the paths1 row and drops every bit already set in half h of the paths0 row
  const int lane = threadIdx.x & 63;
  const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
  for (i64 w = wave; w < count * M; w += nwaves) {
    const i64 i = w / M;
    const int h = (int)(w % M);
    const u32* r0 = p0 + (size_t)row0[i] * S32 + (size_t)h * W32p;
    const u32 r1raw = row1[i];
    const u32 r1 = r1raw & 0x7fffffffu;
    const int h1 = (M == 2 && (r1raw >> 31)) ? 1 - h : h;
    const u64 li = (u64)r1 * M + h1;
    const u64 b1 = loff1[li], e1 = loff1[li + 1];
    u64 out = doff[w];
    const u64 out_end = doff[w + 1];
    for (u64 p = b1; p < e1; p += 64) {
      const u32 e = (p + lane < e1) ? lidx1[p + lane] : zoff;
      const u32 idx = e >> 8;                                  // patient number
      bool keep = (e != zoff);
      const u32 w0 = keep ? r0[idx >> 5] : 0u;
      keep = keep && (((w0 >> (idx & 31u)) & 1u) == 0u);
      const u64 m = __ballot(keep);
      const u32 before = __builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
      if (keep) dlist[out + before] = e;
      out += (u64)__builtin_popcountll(m);
    }
    if (out + lane < out_end) dlist[out + lane] = zoff;       // at most 3 padding entries
  }
}

hipError_t launch_delta_fill(const uint32_t* p0, int S32, int W32p, int method, const uint32_t* row0,
                             const uint32_t* row1, int64_t count, const uint64_t* loff1, const uint32_t* lidx1,
                             const uint64_t* doff, uint32_t zoff, uint32_t* dlist, hipStream_t stream) {
  if (count == 0) return hipSuccess;
  const i64 blocks = (count * method + 3) / 4;
  hipLaunchKernelGGL(k_delta_fill, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, p0, S32,
                     W32p, method, row0, row1, count, loff1, lidx1, doff, zoff, dlist);
  return hipGetLastError();
}

// exclusive scan of u32 counts into u64 offsets: per-1024 partial sums, one block scans them, then local scans
__global__ __launch_bounds__(256) void k_scan_partial(const u32* cnt, i64 n, u64* part) {
  __shared__ u64 red[4];
  const i64 base = (i64)blockIdx.x * 1024;
  u64 s = 0;
  for (int k = threadIdx.x; k < 1024; k += 256)
    if (base + k < n) s += cnt[base + k] & ~3u;   // the low 2 bits are flags, not length
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void k_scan_top(u64* part, i64 nb) {
  // single block: exclusive scan of nb partial sums in place; part[nb] receives the grand total
  __shared__ u64 tmp[1024];
  __shared__ u64 carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (i64 base = 0; base < nb; base += 1024) {
    const i64 i = base + threadIdx.x;
    const u64 v = (i < nb) ? part[i] : 0;
    tmp[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const u64 add = (threadIdx.x >= (unsigned)o) ? tmp[threadIdx.x - o] : 0;
      __syncthreads();
      tmp[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) part[i] = carry + tmp[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += tmp[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[nb] = carry;
}

__global__ __launch_bounds__(256) void k_scan_apply(const u32* cnt, i64 n, const u64* part, u64* off) {
  // one wave-serial pass per 1024 elements: 4 waves x 256 elements, sequential carry inside the block via LDS
  __shared__ u64 wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const i64 base = (i64)blockIdx.x * 1024 + wave * 256;
  u32 v[4];
  u64 s = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const i64 i = base + lane * 4 + k;
    v[k] = (i < n) ? cnt[i] : 0u;
    s += v[k] & ~3u;
  }
  // inclusive scan of the 64 per-lane sums
  u64 incl = s;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u64 t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u64 pre = part[blockIdx.x];
  for (int w = 0; w < wave; w++) pre += wsum[w];
  u64 run = pre + incl - s;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const i64 i = base + lane * 4 + k;
    if (i < n) off[i] = run | (v[k] & 3u);   // flags ride in the low bits of the (4-aligned) offset
    run += v[k] & ~3u;
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) off[n] = part[gridDim.x];
}

hipError_t launch_scan_u32_u64(const uint32_t* cnt, int64_t n, uint64_t* off, uint64_t* scratch, hipStream_t stream) {
  if (n <= 0) return hipMemsetAsync(off, 0, 8, stream);
  const i64 nb = (n + 1023) / 1024;
  hipLaunchKernelGGL(k_scan_partial, dim3((unsigned)nb), dim3(256), 0, stream, cnt, n, scratch);
  hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, stream, scratch, nb);
  hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, stream, cnt, n, scratch, off);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// per-row bit lists of a path set (CSR): entry = (patient index) << 8 = byte offset of the patient's row in a
// mask tile; every row's list is padded with `zoff` (the all-zero row) to a multiple of 4 entries
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 sp_wave_sum(u32 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(256) void k_row_bits(const u32* rows, i64 nrows, int S32, int W32p, u32* cnt) {
  const int lane = threadIdx.x & 63;
  const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
  for (i64 r = wave; r < nrows; r += nwaves) {
    const u32* row = rows + (size_t)r * S32;
    u32 c = 0;
    for (int k = lane; k < W32p; k += 64) c += __builtin_popcount(row[k]);
    c = sp_wave_sum(c);
    if (lane == 0) cnt[r] = (c + 3u) & ~3u;
  }
}

__global__ __launch_bounds__(256) void k_row_fill(const u32* rows, i64 nrows, int S32, int W32p, const u64* off,
                                                  u32 zoff, u32* idx) {
  const int lane = threadIdx.x & 63;
  const i64 wave = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const i64 nwaves = ((i64)gridDim.x * blockDim.x) >> 6;
  const int nch = (W32p + 63) >> 6;
  for (i64 r = wave; r < nrows; r += nwaves) {
    const u32* row = rows + (size_t)r * S32;
    u64 pos = off[r];
    const u64 end = off[r + 1];
    for (int c = 0; c < nch; c++) {
      const int k = c * 64 + lane;
      u32 w = (k < W32p) ? row[k] : 0u;
      const u32 cnt = __builtin_popcount(w);
      const u32 incl = wave_scan_add(cnt);
      u64 p = pos + incl - cnt;
      const u32 base = (u32)c * 2048u + (u32)lane * 32u;
      while (w) {
        const u32 b = __builtin_ctz(w);
        idx[p++] = (base + b) << 8;
        w &= w - 1;
      }
      pos += __builtin_amdgcn_readlane(incl, 63);
    }
    if (pos + lane < end) idx[pos + lane] = zoff;   // at most 3 padding entries
  }
}

hipError_t launch_row_bits(const uint32_t* rows, int64_t nrows, int S32, int W32p, uint32_t* cnt, hipStream_t stream) {
  if (nrows == 0) return hipSuccess;
  const i64 blocks = (nrows + 3) / 4;
  hipLaunchKernelGGL(k_row_bits, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, rows, nrows, S32,
                     W32p, cnt);
  return hipGetLastError();
}

hipError_t launch_row_fill(const uint32_t* rows, int64_t nrows, int S32, int W32p, const uint64_t* off, uint32_t zoff,
                           uint32_t* idx, hipStream_t stream) {
  if (nrows == 0) return hipSuccess;
  const i64 blocks = (nrows + 3) / 4;
  hipLaunchKernelGGL(k_row_fill, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, stream, rows, nrows, S32,
                     W32p, off, zoff, idx);
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
