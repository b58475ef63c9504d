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
#include <cstdlib>

#include "gcre_ie_common.h"

namespace gcre {



// The general kernel: both methods, every count looked up (no pruning).  Runs the signed method, and for method 1 the
// warm-up slice that seeds the pruned kernel's thresholds.  L = counter planes of the joined paths, a multiple of 4.
template <int M, int L>
__global__ __launch_bounds__(64 * kIeWaves) __attribute__((amdgpu_waves_per_eu(M == 1 ? 4 : 2))) void k_null_ie(const IeArgs a) {
  static_assert(L % 4 == 0 && L >= 8 && L <= 16, "planes come in groups of 4");
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int xcd = blockIdx.x & 7;
  const i64 wi = (i64)(blockIdx.x >> 3) * kIeWaves + wave;
  const i64 wx = a.waves_per_xcd;
  const i64 slices = 8 * wx;
  const u32 lane4 = (u32)lane * 4u;
  // the wave's 2048 running maxima of its current tile: 8 KB of LDS, [q][lane], flushed to the global array with
  // atomicMax whenever the wave moves to another tile
  __shared__ u32 gmax_lds[kIeWaves][32 * 64];
  u32* scr = gmax_lds[wave] + lane;
#pragma unroll
  for (int q = 0; q < 32; q++) scr[q * 64] = 0u;

  const SparseSeg GCRE_CONSTANT* segs = (const SparseSeg GCRE_CONSTANT*)a.segs;
  const u64 GCRE_CONSTANT* loff0 = (const u64 GCRE_CONSTANT*)a.loff0;
  const u32 GCRE_CONSTANT* lidx0 = (const u32 GCRE_CONSTANT*)a.lidx0;

  int cur_kt = -1;
  __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)a.mt, 0, 0x7fffffff, 0x00020000);

  auto flush = [&]() {
    if (cur_kt < 0) return;
    u32* out = a.null_bits + (size_t)cur_kt * 2048 + lane * 32;
    for (int q = 0; q < 32; q++) {
      const u32 own = scr[q * 64];
      if (own != 0u) {
        atomicMax(out + q, own);
        scr[q * 64] = 0u;
      }
    }
  };

  auto to_counts = [&](const u32 (&C)[L], u32 (&R)[16]) {
#pragma unroll
    for (int l = 0; l < 16; l++) R[l] = (l < L) ? C[l] : 0u;
    transpose16(R);
  };
  // method 1: counts -> f32 table diagonal -> running maxima (methods.h:96-103).  All 32 table cells and all 32
  // running maxima of the lane are independent loads, in flight together.
  auto finish_m1 = [&](const u32 (&C)[L], u32 total) {
    u32 R[16];
    to_counts(C, R);
    const u32* diag_g = (const u32*)a.t32 + sp_diag_offset(total);
    u32 v[32], o[32];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      v[j] = diag_g[R[j] & 0xffffu];
      v[j + 16] = diag_g[R[j] >> 16];
    }
#pragma unroll
    for (int q = 0; q < 32; q++) o[q] = scr[q * 64];
#pragma unroll
    for (int q = 0; q < 32; q++)
      if (v[q] > o[q]) scr[q * 64] = v[q];
  };

  // method 2: vtmax[a][tp-a] + vtmax[tn-b][b] in f64, rounded to f32, clamped at 0 (methods.h:220-230)
  auto finish_m2 = [&](const u32 (&Cp)[L], const u32 (&Cn)[L], u32 tp, u32 tn) {
    u32 Rp[16], Rn[16];
    to_counts(Cp, Rp);
    to_counts(Cn, Rn);
    const double* dp = a.d64 + sp_diag_offset(tp);
    const double* dn = a.d64 + sp_diag_offset(tn);
#pragma unroll
    for (int g0 = 0; g0 < 16; g0 += 4) {
      double sp[8], sn[8];
      u32 o[8];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        sp[k] = dp[Rp[g0 + k] & 0xffffu];
        sn[k] = dn[Rn[g0 + k] & 0xffffu];
        sp[k + 4] = dp[Rp[g0 + k] >> 16];
        sn[k + 4] = dn[Rn[g0 + k] >> 16];
        o[k] = scr[(g0 + k) * 64];
        o[k + 4] = scr[(g0 + k + 16) * 64];
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        float f = (float)(sp[k] + sn[k]);
        f = (f > 0.0f) ? f : 0.0f;
        const u32 v = __float_as_uint(f);
        const int q = (k < 4) ? g0 + k : g0 + k - 4 + 16;
        if (v > o[k]) scr[q * 64] = v;
      }
    }
  };

  // count planes of one (row-half, tile): `unit` = ((row*M + half) * nkt + tile) * groups; groups of 4 planes,
  // [group][lane][4] dwords = 1 KB per group
  auto load_planes = [&](u32 (&P)[L], const u32* planes, u32 unit, int groups) {
    const u32x4* src = (const u32x4*)(planes + (u64)unit * 256u) + lane;
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
    }
    for (i64 sidx = a.seg_begin + sl; sidx < a.seg_end; sidx += slices) {
      const u32 row0 = segs[sidx].row0;
      const u32 first = segs[sidx].first;
      const u32 npaths = segs[sidx].n;
      // ---- per-path metadata of the whole segment in a few coalesced loads: lane t <-> joined path first + t.
      // The path loop below takes everything out of these registers with v_readlane.
      const u32 qv = first + (((u32)lane < npaths) ? (u32)lane : 0u);
      u32 infov[M], lovv[M];   // padded list length | mode, and where a long list continues
#pragma unroll
      for (int h = 0; h < M; h++) {
        infov[h] = a.linfo[(u64)qv * M + h];
        lovv[h] = a.lover[(u64)qv * M + h];
      }
      const u32 rzv = a.rowz[qv];
      u32 zunit[M];      // where the planes of the added row-half start, in 1 KB units
#pragma unroll
      for (int h = 0; h < M; h++) {
        const u32 hz = (M == 2 && (rzv >> 31)) ? (u32)(1 - h) : (u32)h;
        zunit[h] = ((u32)kt * (u32)a.rowsz + (rzv & 0x7fffffffu) * (u32)M + hz) * (u32)a.gz;
      }
      u32 ttv[M];
#pragma unroll
      for (int h = 0; h < M; h++) ttv[h] = a.tot[(u64)qv * M + h];
      u32 lv[M][8];      // the first 8 entries of every list (lists are padded to 8: most lists end there)
#pragma unroll
      for (int h = 0; h < M; h++) {
        const u32x4* lp = (const u32x4*)(a.dlist + ((u64)qv * M + h) * 8u);
        const u32x4 e0 = lp[0], e1 = lp[1];
        lv[h][0] = e0.x; lv[h][1] = e0.y; lv[h][2] = e0.z; lv[h][3] = e0.w;
        lv[h][4] = e1.x; lv[h][5] = e1.y; lv[h][6] = e1.z; lv[h][7] = e1.w;
      }

      auto stream = [&](u32 (&P)[L], const u32 GCRE_CONSTANT* list, u64 p, u64 e) {
        u32 x[16];
        for (; p + 16 <= e; p += 16) {
          load16(x, mt, lane4, *(const u32x16 GCRE_CONSTANT*)(list + p));
          add16<L>(P, x);
        }
        for (; p < e; p += 4) {
          const u32x4 offs = *(const u32x4 GCRE_CONSTANT*)(list + p);
          u32 y4[4];
#pragma unroll
          for (int j = 0; j < 4; j++) y4[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
          add4<L>(P, y4);
        }
      };

      // ---- base counters: the planes of paths0[row0], or its bits streamed when the set has no planes ----
      u32 B[M][L];
#pragma unroll
      for (int h = 0; h < M; h++) {
        const u64 r = (u64)row0 * M + h;
        if (a.planes0) {
          load_planes(B[h], a.planes0, (u32)(((u64)kt * (u64)a.rows0 + r) * (u64)a.g0), a.g0);
        } else if (a.rec_slot) {
          // the set has the recipe of the join that made it (see k_null_ie_m1 / k_null_ie_m2): per half
          // A[row0'][h] + Z[z'][h'] -/+ that join's list of the half (rec_rows_a / rec_rows_z count row-halves)
          const u32 ra = a.rec_row0[row0], rzr = a.rec_rowz[row0], rz = rzr & 0x7fffffffu, rinfo = a.rec_linfo[r];
          const u32 hz = (M == 2 && (rzr >> 31)) ? (u32)(1 - h) : (u32)h;
          u32 ZR[L], S[L];
          load_planes(B[h], a.rec_planes_a, (u32)(((u64)kt * (u64)a.rec_rows_a + (u64)ra * M + h) * (u64)a.rec_ga), a.rec_ga);
          load_planes(ZR, a.rec_planes_z, (u32)(((u64)kt * (u64)a.rec_rows_z + (u64)rz * M + hz) * (u64)a.rec_gz), a.rec_gz);
#pragma unroll
          for (int l = 0; l < L; l++) S[l] = 0u;
          const u32 rlen = rinfo & kLinfoLenMask;
          stream(S, (const u32 GCRE_CONSTANT*)(a.rec_slot + r * 8u), 0, 8);
          if (rlen > 8u) stream(S, (const u32 GCRE_CONSTANT*)(a.rec_over + a.rec_lover[r]), 0, (u64)(rlen - 8u));
          u32 cy = 0u, bw = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            const u32 zl = (rinfo & 1u) ? ZR[l] : S[l];        // overlap list: + Z - S; delta list: + S
            const u32 sl_ = (rinfo & 1u) ? S[l] : 0u;
            const u32 s1_ = B[h][l] ^ zl ^ cy;
            cy = maj3(B[h][l], zl, cy);
            B[h][l] = s1_ ^ sl_ ^ bw;
            bw = maj3(~s1_, sl_, bw);
          }
        } else {
#pragma unroll
          for (int l = 0; l < L; l++) B[h][l] = 0u;
          stream(B[h], lidx0, loff0[r], loff0[r + 1]);
        }
      }

      for (u32 t = 0; t < npaths; t++) {
        const u32 q = first + t;
        u32 C[M][L];
#pragma unroll
        for (int h = 0; h < M; h++) {
          const u32 r0 = rdlane(infov[h], t);
          const u32 len = r0 & kLinfoLenMask;
          const bool overlap = (r0 & 1u) != 0u;
          // the 8 mask rows every list starts with (zero rows past its real end) ...
          u32 offs[8], y[8];
#pragma unroll
          for (int j = 0; j < 8; j++) offs[j] = rdlane(lv[h][j], t);
#pragma unroll
          for (int j = 0; j < 8; j++) y[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
          // ... and the planes of the row the join adds, when the list is the overlap with paths0
          u32 Z[L];
          if (overlap) {
            load_planes(Z, a.planesz, rdlane(zunit[h], t), a.gz);
          } else {
#pragma unroll
            for (int l = 0; l < L; l++) Z[l] = 0u;
          }
          // sum of the 8 rows: 4 planes
          u32 c0, a0, c1, a1, c2, a2, d0, b0;
          csa(c0, a0, y[0], y[1], y[2]);
          csa(c1, a1, y[3], y[4], y[5]);
          csa(c2, a2, a0, a1, y[6]);
          const u32 s0 = a2 ^ y[7], c3 = a2 & y[7];
          csa(d0, b0, c0, c1, c2);
          const u32 s1 = b0 ^ c3, d1 = b0 & c3;
          u32 S[L];
          S[0] = s0; S[1] = s1; S[2] = d0 ^ d1; S[3] = d0 & d1;
#pragma unroll
          for (int l = 4; l < L; l++) S[l] = 0u;
          if (len > 8u) stream(S, (const u32 GCRE_CONSTANT*)(a.dover + rdlane(lovv[h], t)), 0, (u64)(len - 8u));   // long list (rare)
          if (overlap) {   // C = B + Nz - S
            u32 cy = 0u, bw = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 s1_ = B[h][l] ^ Z[l] ^ cy;
              cy = maj3(B[h][l], Z[l], cy);
              C[h][l] = s1_ ^ S[l] ^ bw;
              bw = maj3(~s1_, S[l], bw);
            }
          } else {         // C = B + S
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              C[h][l] = B[h][l] ^ S[l] ^ cy;
              cy = maj3(B[h][l], S[l], cy);
            }
          }
        }
        if (a.planes_out) {
#pragma unroll
          for (int h = 0; h < M; h++) {
            const u64 rh = ((u64)a.out_first + q) * M + h;
            u32x4* dst = (u32x4*)(a.planes_out + (((u64)cur_kt * (u64)a.rows_out + rh) * (u64)a.go) * 256u) + lane;
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
          finish_m1(C[0], rdlane(ttv[0], t));
        } else {
          finish_m2(C[0], C[M - 1], rdlane(ttv[0], t), rdlane(ttv[M - 1], t));
        }
      }
    }
  }
  flush();
}

// ------------------------------------------------------------------------------------------------
// method 1, the hot kernel.  L counter planes (multiple of 4), GZ plane groups of the added rows, OUT = the joined
// paths' planes are written out (kept joins).  Both operands have planes; lists are padded to 8 entries.
// Per joined path and tile, in the common case: 12 v_readlane, 8 mask-row loads + GZ wide loads, 14 bit ops for the
// 8-row sum, ~3.3 per plane for B + (Nz - S), 2 per plane for the interval test -- every one a single v_bitop3.
// ------------------------------------------------------------------------------------------------

template <int L, int GZ, bool OUT, bool REC>
__global__ __launch_bounds__(64 * kIeWaves) __attribute__((amdgpu_waves_per_eu(L <= 12 ? 4 : 3))) void k_null_ie_m1(const IeArgs a) {
  // L counter planes (any even number: the arithmetic runs over exactly L); plane arrays come in groups of 4
  constexpr int LP = (L + 3) / 4 * 4;
  static_assert(L >= 8 && L <= 16 && GZ >= 2 && 4 * GZ <= LP, "planes come in groups of 4");
  // the waves' running maxima: [q][lane] per wave; touched only by the few lookups that survive the interval test
  __shared__ u32 nmax_lds[kIeWaves][32 * 64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane4 = (u32)lane * 4u;
  u32* nm = nmax_lds[wave] + lane;
#pragma unroll
  for (int q = 0; q < 32; q++) nm[q * 64] = 0u;
  const SparseSeg GCRE_CONSTANT* segs = (const SparseSeg GCRE_CONSTANT*)a.segs;
  // pinned to scalar registers (the compiler re-loads kernel arguments on both sides of the ticket's `lane == 0` branch
  // and then treats them as divergent): the base of the added rows' planes, so that their loads take the scalar-base form
  const char* k_planesz = (const char*)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)((u64)a.planesz >> 32)) << 32) |
                                        (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(u64)a.planesz));

  int cur_kt = -1;
  u32 valid = 0u;
  u32 lad_base = (a.lad_mode == 0) ? 0u : (u32)(kLadderLevels - 1 + a.lad_mode) * (u32)a.ladder_stride;
  const u32 lad_keep = (u32)kLadderLevels * (u32)a.ladder_stride;
  bool dirty = false;     // nm holds maxima the global array has not seen
  u32 n_slow = 0u;
#ifdef GCRE_IE_TIMING
  u64 tm[6] = {0, 0, 0, 0, 0, 0};   // seg prologue, load wait, compute, lookups, exchange, total
  const u64 tm_begin = __builtin_amdgcn_s_memtime();
#define GCRE_TM_MARK(var) __builtin_amdgcn_s_waitcnt(0); const u64 var = __builtin_amdgcn_s_memtime()
#define GCRE_TM_ADD(i, t1, t0) tm[i] += (t1) - (t0)
#else
#define GCRE_TM_MARK(var)
#define GCRE_TM_ADD(i, t1, t0)
#endif
  __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)a.mt, 0, 0x7fffffff, 0x00020000);

  // publish the wave's maxima, read everybody's, set the threshold level to the smallest running maximum of the
  // tile's live permutations (a stale read only lowers it: still exact)
  auto exchange = [&]() {
    // the tile's 2048 running maxima: lane holds 32 consecutive ones = 8 wide loads, issued together.  sc1: served
    // by the memory side, not by this CU's L1 or this XCD's L2, which never see the other XCDs' atomics.
    u32* out = a.null_bits + (size_t)cur_kt * 2048 + lane * 32;
    __amdgpu_buffer_rsrc_t nb = __builtin_amdgcn_make_buffer_rsrc((void*)(a.null_bits + (size_t)cur_kt * 2048), 0, 8192, 0x00020000);
    u32x4 g4[8];
#pragma unroll
    for (int j = 0; j < 8; j++) g4[j] = __builtin_amdgcn_raw_buffer_load_b128(nb, (u32)lane * 128u + (u32)j * 16u, 0, 16 /* sc1 */);
    u32 lo = 0xffffffffu;
#pragma unroll
    for (int q = 0; q < 32; q++) {
      const u32 g = g4[q >> 2][q & 3];
      const u32 own = nm[q * 64];
      if (dirty && own > g) atomicMax(out + q, own);
      const u32 v = own > g ? own : g;
      if ((valid >> q) & 1u) lo = v < lo ? v : lo;
    }
    dirty = false;
    u32 theta = __builtin_amdgcn_readfirstlane(wave_min_u32(lo));
    if (theta == 0xffffffffu) theta = 0u;
    int j = (int)(__uint_as_float(theta) * (float)kLadderPerUnit);   // level j covers thresholds >= j / kLadderPerUnit
    j = j < 0 ? 0 : (j > kLadderLevels - 1 ? kLadderLevels - 1 : j);
    lad_base = (u32)j * (u32)a.ladder_stride;
  };
  auto flush_tile = [&]() {
    if (cur_kt >= 0) {
      u32* out = a.null_bits + (size_t)cur_kt * 2048 + lane * 32;
#pragma unroll 8
      for (int q = 0; q < 32; q++) {
        const u32 own = nm[q * 64];
        if (own != 0u) {
          atomicMax(out + q, own);
          nm[q * 64] = 0u;
        }
      }
    }
    dirty = false;
  };

  // Work queues (WorkQueue above): segments differ in length and tiles in how many counts they look up -- a fixed
  // split of the table leaves the slowest wave 10-40 % behind the average.
  __shared__ u32 wq_state[kIeWaves][8];
  WorkQueue wq;
  wq.st = wq_state[wave];
  wq.init(a.queue, (u32)((a.seg_end - a.seg_begin + a.batch - 1) / a.batch), (u32)a.nkt);
  {
    wq.select(blockIdx.x & 7u);   // workgroups are dealt round the XCDs: blocks b and b + 8 share an L2
  }
  u32 ticket = wq.take(lane);
  int since = 0, period = 1;   // thresholds: refresh after 1, 2, 4, .. segments while they climb, then every kIeRefresh
  for (;;) {
    const u32 work = __builtin_amdgcn_readfirstlane(ticket);
    const u32 q_n = wq.get(2);
    if (work >= q_n) {
      // this queue is empty: take from the fullest one; every wave ends once it has seen them all empty
      if (!wq.steal(lane)) break;
      ticket = wq.take(lane);
      continue;
    }
    ticket = wq.take(lane);   // the next ticket is on its way while this batch is worked on
    const u32 item = wq.get(1) + work, nb = wq.get(3);
    const int kt = (int)(item / nb);
    const i64 s_lo = a.seg_begin + (i64)(item - (u32)kt * nb) * a.batch;
    const i64 s_hi = s_lo + a.batch < a.seg_end ? s_lo + a.batch : a.seg_end;
    if (kt != cur_kt) {
      flush_tile();
      cur_kt = kt;
      mt = __builtin_amdgcn_make_buffer_rsrc((void*)(a.mt + (size_t)kt * a.mt_rows * 64), 0, 0x7fffffff, 0x00020000);
      const int live = a.K - kt * 2048 - lane * 32;
      valid = live >= 32 ? 0xffffffffu : (live <= 0 ? 0u : ((1u << live) - 1u));
      if (a.lad_mode == 0) lad_base = 0u;
      since = 0;
      period = 1;
    }
    // The table entries of a segment -- (row0, first, n) and, with a recipe, the twelve words k_fill_rec_segs gathered
    // for it -- are wave-uniform.  Inside a batch they are fetched one segment ahead into one VGPR (lane l = word l)
    // and moved to scalar registers with v_readlane when their turn comes: the first segment of a batch pays the
    // scalar-load round trip, the others start their plane loads at once.
    u32 nextv = 0u;
    bool have_next = false;
    for (i64 sidx = s_lo; sidx < s_hi; sidx++) {
      u32 row0, first, npaths;
      u32 rc_a = 0u, rc_z = 0u, rc_info = 0u, rc_lov = 0u;
      typedef u32 __attribute__((ext_vector_type(8))) u32x8;
      u32x8 rc_o = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
      if (have_next) {
        row0 = rdlane(nextv, 0);
        first = rdlane(nextv, 1);
        npaths = rdlane(nextv, 2);
        if constexpr (REC) {
          rc_a = rdlane(nextv, 4);
          rc_z = rdlane(nextv, 5);
          rc_info = rdlane(nextv, 6);
          rc_lov = rdlane(nextv, 7);
#pragma unroll
          for (int j = 0; j < 8; j++) rc_o[j] = rdlane(nextv, 8 + j);
        }
      } else {
        row0 = segs[sidx].row0;
        first = segs[sidx].first;
        npaths = segs[sidx].n;
        if constexpr (REC) {
          const u32 GCRE_CONSTANT* rs = (const u32 GCRE_CONSTANT*)a.rec_segs + (u64)sidx * kRecSegWords;
          rc_a = rs[0];
          rc_z = rs[1];
          rc_info = rs[2];
          rc_lov = rs[3];
          rc_o = *(const u32x8 GCRE_CONSTANT*)(rs + 4);
        }
      }
      have_next = sidx + 1 < s_hi;
      if (have_next) {
        const u32* src = (const u32*)a.segs + (u64)(sidx + 1) * 3u + (u32)(lane < 3 ? lane : 0);
        if constexpr (REC) {
          if (lane >= 4 && lane < 16) src = a.rec_segs + (u64)(sidx + 1) * kRecSegWords + (u32)(lane - 4);
        }
        nextv = *src;
      }
      GCRE_TM_MARK(ts0);
      if (a.lad_mode == 0 && ++since >= period) {
        exchange();
        since = 0;
        period = period < kIeRefresh ? period * 2 : kIeRefresh;
      }
      GCRE_TM_MARK(ts1);
      GCRE_TM_ADD(4, ts1, ts0);

      // ---- per-path metadata of the whole segment in a few coalesced loads: lane t <-> joined path first + t ----
      const u32 qv = first + (((u32)lane < npaths) ? (u32)lane : 0u);
      const u32 infov = a.linfo[qv];                 // padded list length | mode
      const u32 lovv = a.lover[qv];                  // where a long list continues
      const u32 zunit = ((u32)kt * (u32)a.rowsz + (a.rowz[qv] & 0x7fffffffu)) * (u32)a.gz;   // 1 KB units into planesz
      const u32 totv = a.tot[qv];
      // segments of another shard's rows: the all-inside row of the ladder (no count is looked up), planes only
      const u32 lhv = a.ladder[((u64)sidx < (u64)a.score_segs ? lad_base : lad_keep) + totv];
      // the first 8 entries of every list (lists are padded to 8: most lists end there) are wave-uniform: they come
      // through the scalar cache, one s_load_dwordx8 per path, fetched one path ahead of the loads that use them
      const u32x8 GCRE_CONSTANT* slots = (const u32x8 GCRE_CONSTANT*)(a.dlist + (u64)first * 8u);
      // OUT (every count is needed): the path's 8 mask rows and the added row's planes.  Otherwise only the planes: the
      // rows are fetched by the lanes the bound filter leaves uncertain (see compute)
      auto issue = [&](u32 t2, const u32x8 offs, u32 (&yy)[8], u32 (&ZZ)[4 * GZ]) {
        if constexpr (OUT) {
#pragma unroll
          for (int j = 0; j < 8; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
        }
        // scalar base + 32-bit lane offset: the address never sits in vector registers (a vector address pair that shares
        // registers with a load still in flight costs a full s_waitcnt vmcnt(0) per path)
        __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(k_planesz + (u64)rdlane(zunit, t2) * 1024u), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < GZ; j++) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rz, lane4 * 4u + (u32)j * 1024u, 0, 0);
          ZZ[4 * j + 0] = v.x; ZZ[4 * j + 1] = v.y; ZZ[4 * j + 2] = v.z; ZZ[4 * j + 3] = v.w;
        }
      };
      // the first path's loads do not depend on the base counters: they go out before those are read (and, with a
      // recipe, rebuilt), one round trip earlier
      u32 yA[8], yB[8], ZA[4 * GZ], ZB[4 * GZ];
      const u32 last = npaths - 1u;
      auto at = [&](u32 t2) -> u32 { return t2 < last ? t2 : last; };
      u32x8 oA = slots[0], oB = slots[at(1u)];
      issue(0u, oA, yA, ZA);
      // ---- base counters: the planes of paths0[row0] -- stored, or (REC) rebuilt from the recipe of the join that
      // produced the row: planes of ITS paths0 row + planes of the row it added -/+ its 8-entry list ----
      u32 B[LP];
      auto load_groups = [&](u32 (&P)[LP], const u32* planes, u64 unit, int groups) {
        const u32x4* src = (const u32x4*)(planes + unit * 256u) + lane;
#pragma unroll
        for (int j = 0; j < LP / 4; j++) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (j < groups) v = src[j * 64];
          P[4 * j + 0] = v.x; P[4 * j + 1] = v.y; P[4 * j + 2] = v.z; P[4 * j + 3] = v.w;
        }
      };
      if constexpr (!REC) {
        load_groups(B, a.planes0, ((u64)kt * (u64)a.rows0 + (u64)row0) * (u64)a.g0, a.g0);
      } else {
        const u32 ra = rc_a, rz = rc_z & 0x7fffffffu, rinfo = rc_info;
        const u32x8 ro = rc_o;
        u32 yr[8];
#pragma unroll
        for (int j = 0; j < 8; j++) yr[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, ro[j], 0);
        u32 ZR[LP];
        load_groups(B, a.rec_planes_a, ((u64)kt * (u64)a.rec_rows_a + (u64)ra) * (u64)a.rec_ga, a.rec_ga);
        load_groups(ZR, a.rec_planes_z, ((u64)kt * (u64)a.rec_rows_z + (u64)rz) * (u64)a.rec_gz, a.rec_gz);
        u32 S[L];
        {
          u32 S4[4];
          sum8(yr, S4);
#pragma unroll
          for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
        }
        const u32 rlen = rinfo & kLinfoLenMask;
        if (rlen > 8u) {   // the producing join's list was long: the rest of it, 8 entries at a time
          const u32 GCRE_CONSTANT* more = (const u32 GCRE_CONSTANT*)(a.rec_over + rc_lov);
          for (u32 p = 0u; p + 8u < rlen; p += 8u) {
            const u32x8 o8 = *(const u32x8 GCRE_CONSTANT*)(more + p);
            u32 yy[8], s4[4];
#pragma unroll
            for (int j = 0; j < 8; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, o8[j], 0);
            sum8(yy, s4);
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 sv = S[l];
              const u32 add = (l < 4) ? s4[l < 4 ? l : 0] : 0u;
              S[l] = xor3(sv, add, cy);
              cy = majority(sv, add, cy);
            }
          }
        }
        if (rinfo & 1u) {   // B = A + Z - S
          u32 cy = 0u, bw = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            const u32 s1_ = xor3(B[l], ZR[l], cy);
            cy = majority(B[l], ZR[l], cy);
            B[l] = xor3(s1_, S[l], bw);
            bw = borrow3(s1_, S[l], bw);
          }
        } else {            // B = A + S
          u32 cy = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            const u32 bl = B[l];
            B[l] = xor3(bl, S[l], cy);
            cy = majority(bl, S[l], cy);
          }
        }
      }

      GCRE_TM_MARK(ts2);
      GCRE_TM_ADD(0, ts2, ts1);
      // ---- the joined paths of the segment, software-pipelined one path ahead: the 8 mask rows and the planes of
      // path t+1 are in flight while path t is computed.  Loads are issued unconditionally (the last path is simply
      // requested twice) so that the compiler's counters let the older loads retire without draining the younger.
      auto compute = [&](u32 t, const u32 (&y)[8], const u32 (&Z)[4 * GZ]) {
        GCRE_TM_MARK(tp0);
        const u32 r0 = rdlane(infov, t);
        const u32 len = r0 & kLinfoLenMask;
        const bool overlap = (r0 & 1u) != 0u;
        u32 C[L];
        u32 S4[4];
        sum8(y, S4);
        if (len <= 8u && overlap) {
          // ---- C = B + (Nz - S): the common case.  Nz - S >= 0: the overlap is part of the added row ----
          u32 T[4 * GZ];
          u32 bw = 0u;
#pragma unroll
          for (int l = 0; l < 4 * GZ; l++) {
            if (l < 4) {
              T[l] = xor3(Z[l], S4[l < 4 ? l : 0], bw);
              bw = borrow3(Z[l], S4[l < 4 ? l : 0], bw);
            } else {
              T[l] = Z[l] ^ bw;
              bw = bw & ~Z[l];
            }
          }
          u32 cy = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            if (l < 4 * GZ) {
              C[l] = xor3(B[l], T[l < 4 * GZ ? l : 0], cy);
              cy = majority(B[l], T[l < 4 * GZ ? l : 0], cy);
            } else {
              C[l] = B[l] ^ cy;
              cy = B[l] & cy;
            }
          }
        } else if (len <= 8u) {
          // ---- C = B + S: the join adds at most 8 patients ----
          u32 cy = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            if (l < 4) {
              C[l] = xor3(B[l], S4[l < 4 ? l : 0], cy);
              cy = majority(B[l], S4[l < 4 ? l : 0], cy);
            } else {
              C[l] = B[l] ^ cy;
              cy = B[l] & cy;
            }
          }
        } else {
          // ---- long list (rare): further blocks of 8 entries, each summed and rippled into a full-width S ----
          u32 S[L];
#pragma unroll
          for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
          const u32 GCRE_CONSTANT* more = (const u32 GCRE_CONSTANT*)(a.dover + rdlane(lovv, t));
          for (u32 p = 0u; p + 8u < len; p += 8u) {
            const u32x8 o8 = *(const u32x8 GCRE_CONSTANT*)(more + p);
            u32 yy[8], s4[4];
#pragma unroll
            for (int j = 0; j < 8; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, o8[j], 0);
            sum8(yy, s4);
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 sv = S[l];
              if (l < 4) {
                S[l] = xor3(sv, s4[l < 4 ? l : 0], cy);
                cy = majority(sv, s4[l < 4 ? l : 0], cy);
              } else {
                S[l] = sv ^ cy;
                cy = sv & cy;
              }
            }
          }
          if (overlap) {   // C = B + Nz - S
            u32 cy = 0u, bw = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 zl = (l < 4 * GZ) ? Z[l < 4 * GZ ? l : 0] : 0u;
              const u32 s1_ = xor3(B[l], zl, cy);
              cy = majority(B[l], zl, cy);
              C[l] = xor3(s1_, S[l], bw);
              bw = borrow3(s1_, S[l], bw);
            }
          } else {         // C = B + S
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              C[l] = xor3(B[l], S[l], cy);
              cy = majority(B[l], S[l], cy);
            }
          }
        }
        if constexpr (OUT) {
          const u64 rh = (u64)a.out_first + first + t;
          u32x4* dst = (u32x4*)(a.planes_out + (((u64)kt * (u64)a.rows_out + rh) * (u64)a.go) * 256u) + lane;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (j < a.go) {
              u32x4 v = {0u, 0u, 0u, 0u};
              if (4 * j < L) v = u32x4{C[(4 * j) % L], (4 * j + 1 < L) ? C[(4 * j + 1) % L] : 0u, (4 * j + 2 < L) ? C[(4 * j + 2) % L] : 0u,
                                       (4 * j + 3 < L) ? C[(4 * j + 3) % L] : 0u};
              dst[j * 64] = v;
            }
          }
        }
        // ---- interval test: live permutations whose count lies outside [lo, hi] of the path's diagonal ----
        const u32 lh = rdlane(lhv, t);
        const u32 lo = lh & 0xffffu, hi = lh >> 16;
        u32 blo = 0u, bhi = 0u;
#pragma unroll
        for (int l = 0; l < L; l++) {
          const u32 kl = (u32)__builtin_amdgcn_sbfe((int)lo, l, 1);   // scalar: all ones when the bound has bit l
          const u32 kh = (u32)__builtin_amdgcn_sbfe((int)hi, l, 1);
          blo = borrow3(C[l], kl, blo);    // C - lo borrows  <=>  C < lo
          bhi = borrow3(kh, C[l], bhi);    // hi - C borrows  <=>  C > hi
        }
        u32 m = (blo | bhi) & valid;
        GCRE_TM_MARK(tp2);
        GCRE_TM_ADD(2, tp2, tp0);
        if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) return;
        // ---- the few permutations that can raise a maximum: rebuild each count from the planes, look it up ----
        n_slow++;
        const u32* diag_g = (const u32*)a.t32 + sp_diag_offset(rdlane(totv, t));
        while (m != 0u) {
          // up to 4 permutations per round: their table cells are independent loads in flight together
          u32 bb[4], vv[4];
#pragma unroll
          for (int k = 0; k < 4; k++) {
            bb[k] = m ? (u32)__builtin_ctz(m) : bb[k ? k - 1 : 0];   // exhausted: repeat the last one (max is idempotent)
            m &= m - 1u;
            u32 cnt = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) cnt |= ((C[l] >> bb[k]) & 1u) << l;
            vv[k] = diag_g[cnt];
          }
#pragma unroll
          for (int k = 0; k < 4; k++)
            __hip_atomic_fetch_max(nm + bb[k] * 64, vv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // ds_max_u32
        }
        dirty = true;
        GCRE_TM_MARK(tp3);
        GCRE_TM_ADD(3, tp3, tp2);
      };

      // ---- !OUT: the bound filter.  count = W - S with W = B + Nz known from the planes alone and 0 <= S <= ov, the
      // length of the path's overlap list.  A permutation with lo + ov <= W <= hi has its count inside [lo, hi] whatever
      // its S is: it cannot raise a maximum, and its 8 mask rows need not be read.  Only the lanes that hold a permutation
      // outside that narrower interval fetch the rows (exec-masked loads), finish the count and test it exactly; what
      // still falls outside [lo, hi] is looked up as before.  ~3/4 of the path-tiles end after 2 plane loads and ~45 bit
      // instructions.  (Delta lists -- rare since the inspector prefers overlap lists -- skip the filter.)
      auto compute_f = [&](u32 t, const u32 (&Z)[4 * GZ]) {
        const u32 r0 = rdlane(infov, t);
        const u32 len = r0 & kLinfoLenMask;
        const bool overlap = (r0 & 1u) != 0u;
        const u32 ov = len - (r0 >> 28);          // the list's true length
        const u32 lh = rdlane(lhv, t);
        const u32 lo = lh & 0xffffu, hi = lh >> 16;
        u32 W[L];
        u32 mu = valid;
        if (overlap) {
          u32 cy = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            if (l < 4 * GZ) {
              W[l] = xor3(B[l], Z[l < 4 * GZ ? l : 0], cy);
              cy = majority(B[l], Z[l < 4 * GZ ? l : 0], cy);
            } else {
              W[l] = B[l] ^ cy;
              cy = B[l] & cy;
            }
          }
          u32 lo2 = lo ? lo + ov : 0u;            // counts are never negative: lo = 0 needs no margin
          lo2 = lo2 > 0xffffu ? 0xffffu : lo2;
          u32 blo = 0u, bhi = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            const u32 kl = (u32)__builtin_amdgcn_sbfe((int)lo2, l, 1);
            const u32 kh = (u32)__builtin_amdgcn_sbfe((int)hi, l, 1);
            blo = borrow3(W[l], kl, blo);    // W < lo + ov
            bhi = borrow3(kh, W[l], bhi);    // W > hi
          }
          if ((lo2 >> L) != 0u) blo = 0xffffffffu;   // the margin pushed the bound past the counters' range
          // W = N0 + Nz can pass 2^L although no count of the path does (the planes are sized for the largest carrier total
          // of the joined paths, and W counts the overlap twice): a carry out of the top plane is "above hi", not a small W
          mu = (blo | bhi | cy) & valid;
        } else {
#pragma unroll
          for (int l = 0; l < L; l++) W[l] = B[l];
        }
        if (__builtin_amdgcn_ballot_w64(mu != 0u) == 0ull) return;
#ifdef GCRE_IE_TIMING
        tm[1] += 1;                                                            // path-tiles the filter left uncertain
        tm[3] += (u64)__popcll(__builtin_amdgcn_ballot_w64(mu != 0u));         // ... and the lanes that fetched rows
#endif
        u32 slow = 0u;
        const u32x8 o = slots[t];
        if (mu != 0u) {
          // ---- the uncertain lanes: rows, exact count, exact test ----
          u32 y[8], S4[4];
#pragma unroll
          for (int j = 0; j < 8; j++) y[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, o[j], 0);
          sum8(y, S4);
          u32 S[L];
#pragma unroll
          for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
          if (len > 8u) {   // long list: further blocks of 8 entries
            const u32 GCRE_CONSTANT* more = (const u32 GCRE_CONSTANT*)(a.dover + rdlane(lovv, t));
            for (u32 p = 0u; p + 8u < len; p += 8u) {
              const u32x8 o8 = *(const u32x8 GCRE_CONSTANT*)(more + p);
              u32 yy[8], s4[4];
#pragma unroll
              for (int j = 0; j < 8; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, o8[j], 0);
              sum8(yy, s4);
              u32 cy = 0u;
#pragma unroll
              for (int l = 0; l < L; l++) {
                const u32 sv = S[l];
                if (l < 4) {
                  S[l] = xor3(sv, s4[l < 4 ? l : 0], cy);
                  cy = majority(sv, s4[l < 4 ? l : 0], cy);
                } else {
                  S[l] = sv ^ cy;
                  cy = sv & cy;
                }
              }
            }
          }
          u32 C[L];
          if (overlap) {   // C = W - S
            u32 bw = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              C[l] = xor3(W[l], S[l], bw);
              bw = borrow3(W[l], S[l], bw);
            }
          } else {         // C = B + S
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              C[l] = xor3(W[l], S[l], cy);
              cy = majority(W[l], S[l], cy);
            }
          }
          u32 blo = 0u, bhi = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            const u32 kl = (u32)__builtin_amdgcn_sbfe((int)lo, l, 1);
            const u32 kh = (u32)__builtin_amdgcn_sbfe((int)hi, l, 1);
            blo = borrow3(C[l], kl, blo);
            bhi = borrow3(kh, C[l], bhi);
          }
          u32 m = (blo | bhi) & mu;
          if (m != 0u) {
            slow = 1u;
            const u32* diag_g = (const u32*)a.t32 + sp_diag_offset(rdlane(totv, t));
            while (m != 0u) {
              u32 bb[4], vv[4];
#pragma unroll
              for (int k = 0; k < 4; k++) {
                bb[k] = m ? (u32)__builtin_ctz(m) : bb[k ? k - 1 : 0];
                m &= m - 1u;
                u32 cnt = 0u;
#pragma unroll
                for (int l = 0; l < L; l++) cnt |= ((C[l] >> bb[k]) & 1u) << l;
                vv[k] = diag_g[cnt];
              }
#pragma unroll
              for (int k = 0; k < 4; k++)
                __hip_atomic_fetch_max(nm + bb[k] * 64, vv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // ds_max_u32
            }
          }
        }
        if (__builtin_amdgcn_ballot_w64(slow != 0u) != 0ull) {
          n_slow++;
          dirty = true;
        }
      };

      if constexpr (OUT) {
        for (u32 t = 0; t < npaths; t += 2) {
          issue(at(t + 1), oB, yB, ZB);
          oA = slots[at(t + 2)];
          compute(t, yA, ZA);
          if (t + 1 < npaths) {
            issue(at(t + 2), oA, yA, ZA);
            oB = slots[at(t + 3)];
            compute(t + 1, yB, ZB);
          }
        }
      } else {
        // a filtered path is ~60 instructions: one path of work does not cover a plane load's latency, so the planes are
        // requested TWO paths ahead (three buffers in rotation).  The list entries are only read by the few paths that
        // fetch rows.
        u32 ZC[4 * GZ];
        issue(at(1u), oB, yB, ZB);
        for (u32 t = 0; t < npaths; t += 3) {
          issue(at(t + 2), oA, yA, ZC);
          compute_f(t, ZA);
          if (t + 1 < npaths) {
            issue(at(t + 3), oA, yA, ZA);
            compute_f(t + 1, ZB);
          }
          if (t + 2 < npaths) {
            issue(at(t + 4), oA, yA, ZB);
            compute_f(t + 2, ZC);
          }
        }
      }
    }
  }
  flush_tile();
#ifdef GCRE_IE_TIMING
  tm[5] = __builtin_amdgcn_s_memtime() - tm_begin;
  if (a.timing && lane == 0)
    for (int i = 0; i < 6; i++) atomicAdd((unsigned long long*)a.timing + i, (unsigned long long)tm[i]);
  if (a.timing && lane == 0) atomicMax((unsigned long long*)a.timing + 6, (unsigned long long)tm[5]);
#endif
  if (a.stats && lane == 0 && n_slow) atomicAdd(a.stats, n_slow);
}

// method 2 runs the general kernel; method 1 the specialised one: L from the largest carrier total, GZ = plane groups of
// the added rows that can be non-zero (<= L/4), OUT = planes of the joined paths wanted
#define GCRE_IE_M2(EXPR)                    \
  if (planes <= 8) { EXPR(2, 8); }          \
  else if (planes <= 12) { EXPR(2, 12); }   \
  else { EXPR(2, 16); }

#define GCRE_IE_M1_OR(EXPR, LL, GG)                                                  \
  if (out) { if (rec) { EXPR(LL, GG, true, true); } else { EXPR(LL, GG, true, false); } }   \
  else { if (rec) { EXPR(LL, GG, false, true); } else { EXPR(LL, GG, false, false); } }

#define GCRE_IE_M1(EXPR)                                                     \
  if (planes <= 8) { GCRE_IE_M1_OR(EXPR, 8, 2) }                             \
  else if (planes <= 10) {                                                   \
    if (gz <= 2) { GCRE_IE_M1_OR(EXPR, 10, 2) } else { GCRE_IE_M1_OR(EXPR, 10, 3) }        \
  } else if (planes <= 12) {                                                 \
    if (gz <= 2) { GCRE_IE_M1_OR(EXPR, 12, 2) } else { GCRE_IE_M1_OR(EXPR, 12, 3) }        \
  } else {                                                                   \
    if (gz <= 2) { GCRE_IE_M1_OR(EXPR, 16, 2) }                              \
    else if (gz == 3) { GCRE_IE_M1_OR(EXPR, 16, 3) }                         \
    else { GCRE_IE_M1_OR(EXPR, 16, 4) }                                      \
  }

#define GCRE_IE_GEN(EXPR)                                          \
  if (method == 1) {                                               \
    if (planes <= 8) { EXPR(1, 8); }                               \
    else if (planes <= 12) { EXPR(1, 12); }                        \
    else { EXPR(1, 16); }                                          \
  } else {                                                         \
    GCRE_IE_M2(EXPR)                                               \
  }

// general = true: the general kernel (both methods; method 1 then looks every count up -- the warm-up slice that seeds
// the thresholds, and joins too small for pruning to pay)
// Per segment of a launch whose paths0 rows come with a recipe: the recipe entries of the segment's row, gathered
// next to the segment table (kRecSegWords words: paths0 row of the producing join, row it added, list info, where a
// long list continues, the list's first 8 entries) so that the pruned kernel needs no load that depends on row0.
__global__ __launch_bounds__(256) void k_fill_rec_segs(const SparseSeg* segs, i64 nsegs, const u32* r_row0, const u32* r_rowz,
                                                       const u32* r_linfo, const u32* r_lover, const u32* r_slot, const u32* r_tot,
                                                       u32* out) {
  const i64 t = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  const i64 sidx = t >> 2;
  const int part = (int)(t & 3);   // four threads per segment: the four scalars, then the slot in two halves
  if (sidx >= nsegs) return;
  const u32 row0 = segs[sidx].row0;
  u32* o = out + sidx * kRecSegWords;
  if (part == 0) {
    o[0] = r_row0[row0];
    o[1] = r_rowz[row0];
  } else if (part == 1) {
    u32 info = r_linfo[row0];
    u32 need = 3u;   // plane groups of the row that can be non-zero, minus one: ceil(bits(carriers) / 4) - 1
    if (r_tot) {
      const u32 t = r_tot[row0];
      need = t < 16u ? 0u : (t < 256u ? 1u : (t < 4096u ? 2u : 3u));
    }
    o[2] = info | (need << 1);
    o[3] = r_lover[row0];
  } else {
    const u32x4 v = *(const u32x4*)(r_slot + (u64)row0 * 8u + (part - 2) * 4);
    *(u32x4*)(o + 4 + (part - 2) * 4) = v;
  }
}

hipError_t launch_fill_rec_segs(const SparseSeg* segs, int64_t nsegs, const uint32_t* r_row0, const uint32_t* r_rowz,
                                const uint32_t* r_linfo, const uint32_t* r_lover, const uint32_t* r_slot, const uint32_t* r_tot,
                                uint32_t* out, hipStream_t stream) {
  if (nsegs == 0) return hipSuccess;
  const i64 blocks = (nsegs * 4 + 255) / 256;
  hipLaunchKernelGGL(k_fill_rec_segs, dim3((unsigned)blocks), dim3(256), 0, stream, segs, nsegs, r_row0, r_rowz, r_linfo, r_lover,
                     r_slot, r_tot, out);
  return hipGetLastError();
}

hipError_t launch_null_ie(const IeArgs& a, int method, int planes, bool general, hipStream_t stream) {
  const dim3 grid((unsigned)(8 * a.waves_per_xcd / kIeWaves));
  const dim3 block(64 * kIeWaves);
  if (method == 1 && !general) {
    const int gz = a.gz;
    const bool out = a.planes_out != nullptr;
    const bool rec = a.rec_slot != nullptr;
#define GCRE_LAUNCH(LL, GG, OO, RR) hipLaunchKernelGGL((k_null_ie_m1<LL, GG, OO, RR>), grid, block, 0, stream, a)
    GCRE_IE_M1(GCRE_LAUNCH)
#undef GCRE_LAUNCH
  } else if (method == 2 && !general) {
    return launch_null_ie_m2(a, planes, stream);
  } else {
#define GCRE_LAUNCH(MM, LL) hipLaunchKernelGGL((k_null_ie<MM, LL>), grid, block, 0, stream, a)
    GCRE_IE_GEN(GCRE_LAUNCH)
#undef GCRE_LAUNCH
  }
  return hipGetLastError();
}

int ie_max_waves_per_cu(int method, int planes, int gz, bool out, bool rec) {
  int blocks = 0;
  hipError_t e = hipSuccess;
  if (method == 1) {
#define GCRE_OCC(LL, GG, OO, RR) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_null_ie_m1<LL, GG, OO, RR>, 64 * kIeWaves, 0)
    GCRE_IE_M1(GCRE_OCC)
#undef GCRE_OCC
  } else {
    return ie2_max_waves_per_cu(planes, gz, out, rec);
  }
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
    u32x4* dst = (u32x4*)(planes + (((u64)w * (u64)nrowhalves + (u64)rr) * (u64)groups) * 256u) + lane;
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
// One wave per diagonal.  A cell c with value d[c] blocks the levels j with j/kLadderPerUnit < d[c], i.e. j < J(c) =
// ceil(kLadderPerUnit * d[c]); the interval of level j ends next to the nearest blocking cell on either side of the
// diagonal's (first) minimum c0:  lo_j = 1 + max{c < c0 : J(c) > j},  hi_j = min{c > c0 : J(c) > j} - 1.
__global__ __launch_bounds__(64) void k_build_ladder(const u32* t32, int TD, u32* ladder) {
  __shared__ int left[kLadderLevels + 2], right[kLadderLevels + 2];
  const int t = blockIdx.x;
  const int lane = threadIdx.x;
  if (t >= TD) return;
  const u32* d = t32 + sp_diag_offset((u32)t);
  for (int k = lane; k < kLadderLevels + 2; k += 64) {
    left[k] = -1;
    right[k] = t + 1;
  }
  // first minimum of the diagonal (values are bit patterns of non-negative floats: they order like the floats)
  u64 mine = ~(u64)0;
  for (int c = lane; c <= t; c += 64) {
    const u64 key = ((u64)d[c] << 32) | (u32)c;
    mine = key < mine ? key : mine;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const u64 other = ((u64)(u32)__shfl_xor((int)(mine >> 32), o, 64) << 32) | (u32)__shfl_xor((int)(u32)mine, o, 64);
    mine = other < mine ? other : mine;
  }
  const u32 best = (u32)(mine >> 32);
  const int c0 = (int)(u32)mine;
  __syncthreads();
  auto blocked_below = [](u32 bits) -> int {   // J: number of levels whose threshold is below the value
    const float v = __uint_as_float(bits);
    if (!(v > 0.0f)) return 0;
    const float x = v * (float)kLadderPerUnit;
    if (x >= (float)kLadderLevels) return kLadderLevels;
    const int j = (int)x;                        // thresholds j/8 are exact in f32, and so is x = 8 v
    return ((float)j < x) ? j + 1 : j;
  };
  for (int c = lane; c <= t; c += 64) {
    if (c == c0) continue;
    const int J = blocked_below(d[c]);
    if (J == 0) continue;
    if (c < c0) atomicMax(&left[J], c);
    else atomicMin(&right[J], c);
  }
  __syncthreads();
  if (lane == 0) {   // suffix max / min over J: a cell with J blocks every level below J
    int l = -1, r = t + 1;
    for (int k = kLadderLevels; k >= 0; k--) {
      l = left[k] > l ? left[k] : l;
      r = right[k] < r ? right[k] : r;
      left[k] = l;
      right[k] = r;
    }
  }
  __syncthreads();
  const int Jbest = blocked_below(best);
  for (int j = lane; j < kLadderLevels; j += 64) {
    // level j is blocked by cells with J > j: left[j + 1] / right[j + 1] after the suffix pass
    u32 e = 1u;   // lo = 1, hi = 0: empty
    if (Jbest <= j) e = ((u32)(right[j + 1] - 1) << 16) | (u32)(left[j + 1] + 1);
    ladder[(size_t)j * TD + t] = e;
  }
  if (lane == 0) {
    // two constant rows behind the levels: "every count is inside" (nothing is looked up: planes-only launches) and
    // "every count is outside" (everything is looked up: GCRE_IE_PRUNE=0)
    ladder[(size_t)kLadderLevels * TD + t] = 0xffff0000u;
    ladder[(size_t)(kLadderLevels + 1) * TD + t] = 1u;
  }
}

hipError_t launch_build_ladder(const float* t32, int TD, uint32_t* ladder, hipStream_t stream) {
  hipLaunchKernelGGL(k_build_ladder, dim3((unsigned)TD), dim3(64), 0, stream, (const u32*)t32, TD, ladder);
  return hipGetLastError();
}

// The signed method's ladder: same construction on the f64 vtmax diagonals at HALF the threshold of each level
// (row r: the interval on which vtmax <= r / (2 * kLadderPerUnit), kLadder2Levels rows), see k_null_ie_m2.
__global__ __launch_bounds__(64) void k_build_ladder2(const double* dmax, int TD, u32* ladder) {
  __shared__ int left[kLadder2Levels + 2], right[kLadder2Levels + 2];
  const int t = blockIdx.x;
  const int lane = threadIdx.x;
  if (t >= TD) return;
  const double* d = dmax + sp_diag_offset((u32)t);
  for (int k = lane; k < kLadder2Levels + 2; k += 64) {
    left[k] = -1;
    right[k] = t + 1;
  }
  // first minimum of the diagonal
  double bestv = __builtin_inf();
  int bestc = 0x7fffffff;
  for (int c = lane; c <= t; c += 64) {
    const double v = d[c];
    if (v < bestv || (v == bestv && c < bestc) || (bestc == 0x7fffffff)) {
      if (!(v != v)) { bestv = v; bestc = c; }
      else if (bestc == 0x7fffffff) { bestv = __builtin_inf(); bestc = c; }   // NaN cells never count as a minimum
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bestv, o, 64);
    const int oc = __shfl_xor(bestc, o, 64);
    if (ov < bestv || (ov == bestv && oc < bestc)) { bestv = ov; bestc = oc; }
  }
  const int c0 = bestc;
  __syncthreads();
  auto blocked_below = [](double v) -> int {   // number of levels whose half threshold j / 16 is below the value
    if (v != v) return kLadder2Levels;          // NaN: never provably small
    if (!(v > 0.0)) return 0;
    const double x = v * (double)(2 * kLadderPerUnit);
    if (x >= (double)kLadder2Levels) return kLadder2Levels;
    const int j = (int)x;
    return ((double)j < x) ? j + 1 : j;
  };
  for (int c = lane; c <= t; c += 64) {
    if (c == c0) continue;
    const int J = blocked_below(d[c]);
    if (J == 0) continue;
    if (c < c0) atomicMax(&left[J], c);
    else atomicMin(&right[J], c);
  }
  __syncthreads();
  if (lane == 0) {
    int l = -1, r = t + 1;
    for (int k = kLadder2Levels; k >= 0; k--) {
      l = left[k] > l ? left[k] : l;
      r = right[k] < r ? right[k] : r;
      left[k] = l;
      right[k] = r;
    }
  }
  __syncthreads();
  const int Jbest = blocked_below(d[c0]);
  for (int j = lane; j < kLadder2Levels; j += 64) {
    u32 e = 1u;   // lo = 1, hi = 0: empty
    if (Jbest <= j) e = ((u32)(right[j + 1] - 1) << 16) | (u32)(left[j + 1] + 1);
    ladder[(size_t)j * TD + t] = e;
  }
  if (lane == 0) {
    ladder[(size_t)kLadder2Levels * TD + t] = 0xffff0000u;
    ladder[(size_t)(kLadder2Levels + 1) * TD + t] = 1u;
  }
}

hipError_t launch_build_ladder2(const double* dmax, int TD, uint32_t* ladder, hipStream_t stream) {
  hipLaunchKernelGGL(k_build_ladder2, dim3((unsigned)TD), dim3(64), 0, stream, dmax, TD, ladder);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// inspector of the IE form, one pass: real-label statistics and observed score of every joined path (what k_stats
// does, methods.h:73-93), the kept row, the check of the reduced operand, the choice delta / overlap list and the
// list itself.  One wave per joined path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 ie_score_key(double s) {
  // order-preserving image of a double; 0 = "not a candidate" (score not > -inf, or NaN: methods.h:91)
  if (!(s > -__builtin_inf())) return 0;
  const u64 b = (u64)__double_as_longlong(s);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// inclusive prefix sum inside every row of 16 lanes (DPP row_shr: no LDS); lane 15 of a row ends up with its total
__device__ __forceinline__ u32 row_scan_add(u32 v) {
  u32 s = v;
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);   // row_shr:1
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);   // row_shr:2
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);   // row_shr:3
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, true);   // row_shr:4, banks 1-3
  s += (u32)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, true);   // row_shr:8, banks 2-3
  return s;
}
// value of lane 15 of the caller's row, in every lane of the row
__device__ __forceinline__ u32 row_last(u32 v, int lane) {
  return (u32)__builtin_amdgcn_ds_bpermute(((lane | 15) << 2), (int)v);
}
__device__ __forceinline__ u32 row_total(u32 v, int lane) { return row_last(row_scan_add(v), lane); }

// Sixteen lanes per joined path, four paths per wave: a path row is Wp <= 1024 words, a rare-variant cohort has ~80,
// and everything per path (counts, decisions, list positions) stays in vector registers, uniform inside a row.
// NIT > 0: the row fits NIT passes of 16 lanes (Wp <= 16 * NIT): all its words are loaded up front, in flight
// together, and stay in registers for the list pass.  NIT == 0: any width, words are read again for the lists.
template <int M, int NIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((M == 1 && NIT > 0 && NIT <= 5) ? 6 : 4))) void k_stats_ie(const StatsArgs a) {
  constexpr int NW = NIT > 0 ? NIT : 1;
  const int lane = threadIdx.x & 63;
  const int sl = lane & 15;
  const i64 wave = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
  const i64 nwaves = (i64)gridDim.x * 4;
  const int Wp = a.Wp;
  u32 my_max_tot = 0, my_modes = 0, my_max_len = 0;
  bool my_bad = false;
  constexpr u32 kOverChunk = 2048;   // entries of the overflow area a wave reserves at a time
  u32 chunk_at = 0u, chunk_left = 0u;
  // the row numbers of the next group of four paths (two dependent loads) are fetched while this one is worked on
  // (rng: the uid's row of the excess table, or kNoRange when an earlier path of this launch shares the uid -- the
  // check "excess inside paths0[idx]" is per uid, one path of it is enough)
  constexpr u32 kNoRange = 0xffffffffu;
  auto fetch = [&](i64 base, u32& r0, u32& r1raw, u32& zraw, u32& rng) {
    const i64 i = (base + (lane >> 4) < a.count) ? base + (lane >> 4) : a.count - 1;
    r0 = a.row0[i];
    r1raw = a.row1[i];
    zraw = a.zindex ? (u32)a.zindex[r1raw & 0x7fffffffu] : (r1raw & 0x7fffffffu);
    rng = kNoRange;
    if (a.excess && (i == 0 || a.row0[i - 1] != r0)) rng = (u32)a.range_of[r0];
  };
  u32 n_r0 = 0, n_r1raw = 0, n_zraw = 0, n_rng = 0;
  if (wave * 4 < a.count) fetch(wave * 4, n_r0, n_r1raw, n_zraw, n_rng);
  u64 cmw[NW];   // this lane's words of the case mask: the same for every path
#pragma unroll
  for (int it = 0; it < NW; it++) cmw[it] = (NIT > 0 && it * 16 + sl < Wp) ? a.case_mask[it * 16 + sl] : 0;
  for (i64 base = wave * 4; base < a.count; base += nwaves * 4) {
    const bool active = base + (lane >> 4) < a.count;
    const i64 i = active ? base + (lane >> 4) : a.count - 1;   // idle rows shadow the last path and write nothing
    const u32 r0 = n_r0, r1raw = n_r1raw, zraw = n_zraw, rng = n_rng;
    if (base + nwaves * 4 < a.count) fetch(base + nwaves * 4, n_r0, n_r1raw, n_zraw, n_rng);
    const u32 rz = zraw & 0x7fffffffu;
    const u32 zflip = (r1raw ^ (a.zindex ? zraw : 0u)) & 0x80000000u;
    const u64* x = a.p0 + (size_t)r0 * a.S;
    const u64* z = a.pz + (size_t)rz * a.S;
    // The joined row is paths0[idx] | paths1[loc] (methods.h:77-78 / :164-165).  paths1[loc] is never read here:
    // without a hint z IS paths1[loc]; with one, paths1[loc] = z | excess (k_range_union checked z inside it) and
    // the union U of the excess over every row this uid joins must lie inside paths0[idx] -- checked below -- so
    // paths0[idx] | paths1[loc] == paths0[idx] | z for every path of the uid.
    const u64* uu = (rng != kNoRange) ? a.excess + (size_t)rng * a.S : nullptr;
    u64* out = (a.res && active) ? a.res + (size_t)(a.first + i) * a.S : nullptr;
    const bool swap = (M == 2) && (r1raw >> 31) != 0;
    const u64* uh[2] = {(uu && swap) ? uu + Wp : uu, (uu && !swap) ? uu + Wp : uu};
    const u64* zh[2] = {zflip ? z + Wp : z, zflip ? z : z + Wp};

    // ---- pass 1: counts, 16 bits each, two to a register (64 * Wp < 65535) ----
    u32 cc[M], dv[M];          // carriers among cases | among controls << 16, delta | overlap << 16
    u64 xw[M][NW], zw[M][NW];  // NIT > 0: the row's words, kept for the list pass
#pragma unroll
    for (int h = 0; h < M; h++) cc[h] = dv[h] = 0u;
    u64 stray = 0;   // excess bits outside paths0[idx]: the hint does not describe this uid
    auto tally = [&](int h, int k, u64 xk, u64 zk, u64 uk, u64 cm) {
      const u64 j = xk | zk;
      if (out) out[h * Wp + k] = j;
      stray |= uk & ~xk;
      cc[h] += (u32)__popcll(j & cm) | ((u32)__popcll(j) << 16);          // among the cases | all (controls by difference)
      dv[h] += (u32)__popcll(zk & ~xk) | ((u32)__popcll(zk) << 16);       // new | all of z (overlap by difference)
    };
    if constexpr (NIT > 0) {
      u64 uw[M][NW];
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int k = it * 16 + sl;
        const bool in = k < Wp;
#pragma unroll
        for (int h = 0; h < M; h++) {
          xw[h][it] = in ? x[h * Wp + k] : 0;
          zw[h][it] = in ? zh[h][k] : 0;
          uw[h][it] = (in && uu) ? uh[h][k] : 0;
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int k = it * 16 + sl;
        if (k < Wp) {
#pragma unroll
          for (int h = 0; h < M; h++) tally(h, k, xw[h][it], zw[h][it], uw[h][it], cmw[it]);
        }
      }
    } else {
      for (int k = sl; k < Wp; k += 16) {
        const u64 cm = a.case_mask[k];
#pragma unroll
        for (int h = 0; h < M; h++) tally(h, k, x[h * Wp + k], zh[h][k], uu ? uh[h][k] : 0, cm);
      }
    }
    if (stray) my_bad = true;
    u32 tot[M], mode[M], len[M], inc[M], inm[M];
#pragma unroll
    for (int h = 0; h < M; h++) {
      const u32 c = row_total(cc[h], lane), d = row_total(dv[h], lane);
      inc[h] = c & 0xffffu;      // carriers among the cases
      tot[h] = c >> 16;
      inm[h] = tot[h] - inc[h];  // carriers among the controls
      const u32 dl = d & 0xffffu, ov = (d >> 16) - dl;
      mode[h] = a.ie_rule ? ((ov <= 8u || ov < dl) ? 1u : 0u) : ((a.ie_bias >= 0 && ov + (u32)a.ie_bias < dl) ? 1u : 0u);
      len[h] = mode[h] ? ov : dl;
      if (active && sl == 0) {
        my_modes += mode[h];
        my_max_tot = max(my_max_tot, tot[h]);
      }
    }
    double score = 0.0;   // looked up here, stored after the lists are out: the wave does not sit on the (cold) table load
    if (active && sl == 0) {
      if constexpr (M == 1) {
        score = a.dvt[(size_t)sp_diag_offset(tot[0]) + inc[0]];   // vt[cases][ctrls], methods.h:90
        a.tot[i] = tot[0];
        a.cases[i] = inc[0];
        a.ctrls[i] = inm[0];
      } else {
        // (+) half: case_pos = inc[0], ctrl_neg = inm[0]; (-) half: ctrl_pos = inc[1], case_neg = inm[1] (methods.h:182-185)
        const u32 case_pos = inc[0], ctrl_neg = inm[0], ctrl_pos = inc[M - 1], case_neg = inm[M - 1];
        score = a.dvt[(size_t)sp_diag_offset(tot[0]) + case_pos] + a.dvt[(size_t)sp_diag_offset(tot[M - 1]) + case_neg];
        a.tot[2 * i] = tot[0];
        a.tot[2 * i + 1] = tot[M - 1];
        a.cases[i] = case_pos + case_neg;        // methods.h:256-257
        a.ctrls[i] = ctrl_pos + ctrl_neg;
      }
      a.rowz[i] = rz | zflip;
    }

    // ---- pass 2: the lists.  Entry = patient << 8 (byte offset of the patient's row in a mask tile) ----
#pragma unroll
    for (int h = 0; h < M; h++) {
      const u64 d = (u64)i * M + h;
      const u32 len8 = max(8u, (len[h] + 7u) & ~7u);
      // long lists keep their tail in the overflow area.  Every wave carves it out of a private chunk and only goes
      // to the shared counter for a new chunk (one same-address atomic with return per list costs microseconds)
      const u32 need = (active && len8 > 8u) ? len8 - 8u : 0u;
      const u32 n0 = rdlane(need, 0), n1 = rdlane(need, 16), n2 = rdlane(need, 32), n3 = rdlane(need, 48);
      const u32 nsum = n0 + n1 + n2 + n3;
      u32 ovb = 0u;
      if (nsum != 0u) {
        if (chunk_left < nsum) {
          const u32 grab = nsum > kOverChunk ? nsum : kOverChunk;
          u32 wbase = 0u;
          if (lane == 0) wbase = atomicAdd(a.ov_count, grab);
          chunk_at = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
          chunk_left = grab;
        }
        const int row = lane >> 4;
        ovb = chunk_at + (row > 0 ? n0 : 0u) + (row > 1 ? n1 : 0u) + (row > 2 ? n2 : 0u);
        chunk_at += nsum;
        chunk_left -= nsum;
      }
      const bool ov_ok = active && len8 > 8u && (u64)ovb + (len8 - 8u) <= (u64)a.over_cap;
      u32* slot = a.slot + d * 8;
      u32* over = a.over + ovb;
      // positions: lane-major inside the row (any order of a list is as good as any other): one row scan of the
      // lanes' entry counts, then every lane writes its own entries back to back
      auto word = [&](int it) -> u64 {
        if constexpr (NIT > 0) {
          return mode[h] ? (zw[h][it] & xw[h][it]) : (zw[h][it] & ~xw[h][it]);
        } else {
          const int k = it * 16 + sl;
          if (k >= Wp) return 0;
          const u64 xk = x[h * Wp + k], zk = zh[h][k];
          return mode[h] ? (zk & xk) : (zk & ~xk);
        }
      };
      const int nit = NIT > 0 ? NIT : (Wp + 15) / 16;
      u32 mine = 0u;
      if constexpr (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NIT; it++) mine += (u32)__popcll(word(it));
      } else {
        for (int it = 0; it < nit; it++) mine += (u32)__popcll(word(it));
      }
      u32 pos = row_scan_add(mine) - mine;
      auto emit = [&](int it) {
        u64 w = word(it);
        const u32 k = (u32)(it * 16 + sl);
        while (w) {
          const u32 b = (u32)__builtin_ctzll(w);
          w &= w - 1;
          const u32 e = (k * 64u + b) << 8;
          if (pos < 8u) { if (active) slot[pos] = e; }
          else if (ov_ok) over[pos - 8u] = e;
          pos++;
        }
      };
      if constexpr (NIT > 0) {
#pragma unroll
        for (int it = 0; it < NIT; it++) emit(it);
      } else {
        for (int it = 0; it < nit; it++) emit(it);
      }
      for (u32 p = len[h] + (u32)sl; p < len8; p += 16) {   // padding: the all-zero mask row
        if (p < 8u) { if (active) slot[p] = a.zoff; }
        else if (ov_ok) over[p - 8u] = a.zoff;
      }
      if (active && sl == 0) {
        a.linfo[d] = len8 | mode[h] | ((len8 - len[h]) << 28);
        a.lover[d] = ovb;
        my_max_len = max(my_max_len, len8);
      }
    }
    if (active && sl == 0) a.key[i] = ie_score_key(score);
  }
  if (my_max_tot) atomicMax(a.max_tot, my_max_tot);
  if (my_bad) *a.bad = 1u;
  if (my_modes) atomicAdd(a.bad + 1, my_modes);   // statistics: overlap-mode lists
  if (my_max_len > 8u) atomicMax(a.max_tot + 5, my_max_len);   // longest list (padded): the quad kernel sums up to 56 entries
}

// Excess of paths1 over the reduced operand, per distinct (location, count) range of the join index:
//   U[r] = OR over loc in the range of  paths1[loc] & ~z'(loc),   z'(loc) = reduced[index[loc]] in paths1's orientation
// and the check that z'(loc) lies inside paths1[loc].  One wave per (range, row) pair, OR-ed into U with atomics (the
// excess is the pivot gene's carriers: a few dozen non-zero words); U is zero on entry.
__global__ __launch_bounds__(256) void k_range_union(const u64* p1, const u64* pz, const int32_t* zindex, const int32_t* pair_range,
                                                     const i64* pair_loc, i64 npairs, int S, int Wp, int M, u64* excess,
                                                     u32* bad) {
  const int lane = threadIdx.x & 63;
  const i64 wave = (i64)blockIdx.x * 4 + (threadIdx.x >> 6);
  const i64 nwaves = (i64)gridDim.x * 4;
  bool my_bad = false;
  for (i64 p = wave; p < npairs; p += nwaves) {
    const i64 loc = pair_loc[p];
    const u32 zraw = (u32)zindex[loc];
    const u64* zrow = pz + (size_t)(zraw & 0x7fffffffu) * S;
    const u64* yrow = p1 + (size_t)loc * S;
    u64* urow = excess + (size_t)pair_range[p] * S;
    for (int w = lane; w < S; w += 64) {
      const int h = w / Wp, k = w - h * Wp;
      const int hz = (M == 2 && (zraw >> 31)) ? 1 - h : h;
      const u64 zw = zrow[(size_t)hz * Wp + k];
      const u64 yw = yrow[w];
      if (zw & ~yw) my_bad = true;
      const u64 e = yw & ~zw;
      if (e) atomicOr((unsigned long long*)(urow + w), (unsigned long long)e);
    }
  }
  if (my_bad) *bad = 1u;
}

hipError_t launch_range_union(const uint64_t* p1, const uint64_t* pz, const int32_t* zindex, const int32_t* pair_range,
                              const int64_t* pair_loc, int64_t npairs, int S, int Wp, int method, uint64_t* excess,
                              uint32_t* bad, hipStream_t stream) {
  if (npairs == 0) return hipSuccess;
  const i64 blocks = (npairs + 3) / 4;
  hipLaunchKernelGGL(k_range_union, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, stream, p1, pz, zindex,
                     pair_range, pair_loc, npairs, S, Wp, method, excess, bad);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The method-1 inspector, block-staged form.  Same outputs as k_stats_ie<1, .>, bit for bit, but organised around what
// bounded that kernel: not its ~110 VALU instructions per path but its ~7.5 vector-memory instructions per path (the CU's
// vector-memory pipe takes ~15 clocks per wave-level load or store whatever its width, tools/row_gather_rate.hip) -- seven
// 4-byte stores per path by one lane in sixteen, one store per list entry, 8-byte row loads.  Here a wave owns 64
// CONSECUTIVE joined paths: their row numbers are read with one coalesced load per array, their seven result words and
// their 8-entry list slots are collected in LDS and written with one coalesced store per array per 64 paths, and the
// rows are read 16 bytes per lane.  Sixteen lanes per path, four paths at a time, as before.
// NL = 16-byte loads per row and lane = ceil(Wp / 32) (Wp <= 32 * NL).
// ------------------------------------------------------------------------------------------------
#ifndef GCRE_STATS_WAVES
#define GCRE_STATS_WAVES 4   // 122 VGPRs with one set of row registers (below): four waves per SIMD without a spill; 6.0 against 6.7 ms per
                             // pass at three.  Five (96 VGPRs, 27 spilled, the pair buffer flushed in chunks to fit the LDS): 8.1 ms
#endif
template <int NL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NL <= 3 ? GCRE_STATS_WAVES : 2))) void k_stats_ie2(const StatsArgs a) {
  typedef u64 __attribute__((ext_vector_type(2))) u64x2;
  constexpr u32 kNoRange = 0xffffffffu;
  constexpr u32 kOverChunk = 2048;
  __shared__ u32 slot_lds[4][64 * 8];   // the 64 paths' list slots
  __shared__ u32 out_lds[4][8][64];     // their result words: tot, cases, ctrls, rowz, key lo, key hi, linfo, lover
  __shared__ u32 pair_lds[4][4][3][32 * NL];   // per group: the non-zero words of the current path's list (low, high, word index)
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sl = lane & 15, grp = lane >> 4;
  const u32 gsh = (u32)grp * 16u, ltm = (1u << sl) - 1u;
  u32 (*pairs)[32 * NL] = pair_lds[wv][grp];
  __shared__ u32 meta_lds[4][3][64];    // the 64 paths' paths0 row, reduced row, range
  u32 (*meta)[64] = meta_lds[wv];
  const i64 wave = (i64)blockIdx.x * 4 + wv;
  const i64 nwaves = (i64)gridDim.x * 4;
  const int Wp = a.Wp;
  u32* slots = slot_lds[wv];
  u32 (*outs)[64] = out_lds[wv];
  u32 my_max_tot = 0, my_modes = 0, my_max_len = 0;
  bool my_bad = false;
  u32 chunk_at = 0u, chunk_left = 0u;
  // this lane's words of a row: 2 sl, 2 sl + 1 of every 32-word block
  // the case mask sits in LDS (words beyond Wp zero), read where it is used: twelve registers less per lane
  __shared__ u64 cm_lds[32 * NL];
  for (int k = (int)threadIdx.x; k < 32 * NL; k += 256) cm_lds[k] = k < Wp ? a.case_mask[k] : 0;
  __syncthreads();
  const i64 nblocks = (a.count + 63) / 64;
  for (i64 blk = wave; blk < nblocks; blk += nwaves) {
    const i64 base = blk * 64;
    const i64 iq = base + lane < a.count ? base + lane : a.count - 1;
    // ---- the 64 paths' row numbers, one coalesced load per array (lane t <-> path base + t) ----
    const u32 r0v = a.row0[iq];
    const u32 r1v = a.row1[iq];
    const u32 zv = a.zindex ? (u32)a.zindex[r1v & 0x7fffffffu] : (r1v & 0x7fffffffu);
    u32 rngv = kNoRange;   // the uid's row of the excess table, for the first path of a uid in this launch only
    if (a.excess && (iq == 0 || a.row0[iq - 1] != r0v)) rngv = (u32)a.range_of[r0v];
    // (lane t <-> path t: parked in LDS, read back per group -- three registers less than keeping them for ds_bpermute)
    meta[0][lane] = r0v;
    meta[1][lane] = zv & 0x7fffffffu;
    meta[2][lane] = rngv;
    __builtin_amdgcn_wave_barrier();
    // slots start as padding
    {
      const u32x4 pad = {a.zoff, a.zoff, a.zoff, a.zoff};
      ((u32x4*)slots)[lane * 2] = pad;
      ((u32x4*)slots)[lane * 2 + 1] = pad;
    }
    // the observed score is a gather from a 100-MB table (a miss all the way to HBM): its key is written one path late,
    // so that the wave never sits on that load
    double score_prev = 0.0;
    int pl_prev = -1;
    // The rows of the next four paths are requested as soon as these four have given up their list words (behind the
    // collection pass below), straight into the registers the current rows occupied: one set of row registers instead of
    // two is what lets a fourth wave onto the SIMD, and the other waves cover what the shorter distance no longer does.
    // (consecutive joined paths mostly share their paths0 row -- one uid joins ~11 rows at level 4: a group that stays
    // on the same row keeps its words, and when no group moves on the load is not issued at all)
    u64 xw[NL][2], zw[NL][2];
#pragma unroll
    for (int it = 0; it < NL; it++) xw[it][0] = xw[it][1] = zw[it][0] = zw[it][1] = 0ull;
    u32 n_rz = 0u, n_rng = kNoRange, n_r0 = 0xffffffffu;
    auto fetch_rows = [&](int it4n) {
      const int pln = it4n * 4 + grp;
      const u32 r0n = meta[0][pln];
      n_rz = meta[1][pln];
      n_rng = meta[2][pln];
      const bool new_x = r0n != n_r0;
      n_r0 = r0n;
      const u64* xn = a.p0 + (size_t)r0n * a.S;
      const u64* zn = a.pz + (size_t)n_rz * a.S;
#pragma unroll
      for (int it = 0; it < NL; it++) {
        const int k = it * 32 + 2 * sl;
        if (k < Wp) {            // Wp is a multiple of 4: words k and k + 1 are both inside
          if (new_x) {
            const u64x2 v = *(const u64x2*)(xn + k);
            xw[it][0] = v.x; xw[it][1] = v.y;
          }
          const u64x2 v = *(const u64x2*)(zn + k);
          zw[it][0] = v.x; zw[it][1] = v.y;
        }
      }
    };
    fetch_rows(0);
    for (int it4 = 0; it4 < 16; it4++) {
      const int pl = it4 * 4 + grp;                 // the group's path inside the block
      const bool active = base + pl < a.count;
      if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
      const u32 rz = n_rz, rng = n_rng;
      const u64* uu = (rng != kNoRange) ? a.excess + (size_t)rng * a.S : nullptr;
      u64* out = (a.res && active) ? a.res + (size_t)(a.first + base + pl) * a.S : nullptr;
      u32 cc = 0u, dv = 0u;
      u64 stray = 0;
#pragma unroll
      for (int it = 0; it < NL; it++) {
        const int k = it * 32 + 2 * sl;
        u64x2 uv = {0, 0};
        if (k < Wp && uu) uv = *(const u64x2*)(uu + k);
        const u64x2 cmv = *(const u64x2*)(cm_lds + k);
        const u64 cme[2] = {cmv.x, cmv.y};
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const u64 xk = xw[it][e], zk = zw[it][e], uk = e ? uv.y : uv.x;
          stray |= uk & ~xk;
          const u64 j = xk | zk;
          cc += (u32)__popcll(j & cme[e]) | ((u32)__popcll(j) << 16);
          dv += (u32)__popcll(zk & ~xk) | ((u32)__popcll(zk) << 16);
        }
        if (out && k < Wp) *(u64x2*)(out + k) = u64x2{xw[it][0] | zw[it][0], xw[it][1] | zw[it][1]};
      }
      if (stray) my_bad = true;
      const u32 c = row_total(cc, lane), d = row_total(dv, lane);
      const u32 inc = c & 0xffffu, tot = c >> 16, inm = tot - inc;
      const u32 dl = d & 0xffffu, ov = (d >> 16) - dl;
      const u32 mode = a.ie_rule ? ((ov <= 8u || ov < dl) ? 1u : 0u) : ((a.ie_bias >= 0 && ov + (u32)a.ie_bias < dl) ? 1u : 0u);
      const u32 len = mode ? ov : dl;
      const u32 len8 = max(8u, (len + 7u) & ~7u);
      // The list's bits are few and scattered -- three overlapping patients among a path's 96 lane-words -- so a loop per
      // word runs its body for one lane at a time.  Instead the non-zero words are first collected per group (word, its
      // index; positions from a ballot), then every lane takes one collected word and all of them give up a bit per round:
      // two rounds instead of eight bodies.  Entries come out in collection order, not ascending (the list is a set).
      u32 npair = 0u;
#pragma unroll
      for (int it = 0; it < NL; it++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const u64 w = mode ? (zw[it][e] & xw[it][e]) : (zw[it][e] & ~xw[it][e]);
          const bool nz = active && w != 0;
          const u64 bal = __builtin_amdgcn_ballot_w64(nz);
          if (bal == 0ull) continue;
          const u32 m = (u32)(bal >> gsh) & 0xffffu;
          const u32 at = npair + (u32)__builtin_popcount(m & ltm);
          if (nz) {
            pairs[0][at] = (u32)w;
            pairs[1][at] = (u32)(w >> 32);
            pairs[2][at] = (u32)(it * 32 + 2 * sl + e);
          }
          npair += (u32)__builtin_popcount(m);
        }
      }
      fetch_rows(it4 < 15 ? it4 + 1 : 15);   // (changes rz / rng of the NEXT iteration only: this one read them above)
      double score = 0.0;
      if (active && sl == 0) {
        my_modes += mode;
        my_max_tot = max(my_max_tot, tot);
        my_max_len = max(my_max_len, len8);
        score = a.dvt[(size_t)sp_diag_offset(tot) + inc];   // vt[cases][ctrls], methods.h:90
      }
      // ---- the list: first 8 entries into the LDS slot, the rest into the overflow area (chunk reserved per wave) ----
      const u32 need = (active && len8 > 8u) ? len8 - 8u : 0u;
      const u32 n0 = rdlane(need, 0), n1 = rdlane(need, 16), n2 = rdlane(need, 32), n3 = rdlane(need, 48);
      const u32 nsum = n0 + n1 + n2 + n3;
      u32 ovb = 0u;
      if (nsum != 0u) {
        if (chunk_left < nsum) {
          const u32 grab = nsum > kOverChunk ? nsum : kOverChunk;
          u32 wbase = 0u;
          if (lane == 0) wbase = atomicAdd(a.ov_count, grab);
          chunk_at = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
          chunk_left = grab;
        }
        ovb = chunk_at + (grp > 0 ? n0 : 0u) + (grp > 1 ? n1 : 0u) + (grp > 2 ? n2 : 0u);
        chunk_at += nsum;
        chunk_left -= nsum;
      }
      const bool ov_ok = active && len8 > 8u && (u64)ovb + (len8 - 8u) <= (u64)a.over_cap;
      u32* over = a.over + ovb;
      __builtin_amdgcn_wave_barrier();
      {
        const u32 npmax = max(max(rdlane(npair, 0), rdlane(npair, 16)), max(rdlane(npair, 32), rdlane(npair, 48)));
        u32 cnt = 0u;
        for (u32 p0 = 0u; p0 < npmax; p0 += 16u) {
          const u32 pi = p0 + (u32)sl;
          const bool has = pi < npair;
          u64 w = has ? ((u64)pairs[1][pi] << 32) | (u64)pairs[0][pi] : 0ull;
          const u32 k = has ? pairs[2][pi] : 0u;
          while (__builtin_amdgcn_ballot_w64(w != 0ull) != 0ull) {
            const bool nzb = w != 0ull;
            const u32 m = (u32)(__builtin_amdgcn_ballot_w64(nzb) >> gsh) & 0xffffu;
            const u32 pos = cnt + (u32)__builtin_popcount(m & ltm);
            const u32 b = nzb ? (u32)__builtin_ctzll(w) : 0u;
            w &= w - 1ull;
            const u32 en = (k * 64u + b) << 8;
            if (nzb) {
              if (pos < 8u) slots[pl * 8 + (int)pos] = en;
              else if (ov_ok) over[pos - 8u] = en;
            }
            cnt += (u32)__builtin_popcount(m);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (u32 p = max(len, 8u) + (u32)sl; p < len8; p += 16)   // padding of the overflow part
        if (ov_ok) over[p - 8u] = a.zoff;
      // ---- the path's result words into LDS, lane pl of every array ----
      if (pl_prev >= 0) {
        const u64 key = ie_score_key(score_prev);
        outs[4][pl_prev] = (u32)key;
        outs[5][pl_prev] = (u32)(key >> 32);
      }
      pl_prev = -1;
      if (active && sl == 0) {
        outs[0][pl] = tot;
        outs[1][pl] = inc;
        outs[2][pl] = inm;
        outs[3][pl] = rz;
        outs[6][pl] = len8 | mode | ((len8 - len) << 28);
        outs[7][pl] = ovb;
        score_prev = score;
        pl_prev = pl;
      }
    }
    if (pl_prev >= 0) {
      const u64 key = ie_score_key(score_prev);
      outs[4][pl_prev] = (u32)key;
      outs[5][pl_prev] = (u32)(key >> 32);
    }
    // ---- 64 paths' results and slots, one coalesced store per array ----
    if (base + lane < a.count) {
      const i64 i = base + lane;
      a.tot[i] = outs[0][lane];
      a.cases[i] = outs[1][lane];
      a.ctrls[i] = outs[2][lane];
      a.rowz[i] = outs[3][lane];
      a.key[i] = ((u64)outs[5][lane] << 32) | outs[4][lane];
      a.linfo[i] = outs[6][lane];
      a.lover[i] = outs[7][lane];
      u32x4* dst = (u32x4*)(a.slot + (u64)i * 8u);
      dst[0] = ((const u32x4*)slots)[lane * 2];
      dst[1] = ((const u32x4*)slots)[lane * 2 + 1];
    }
  }
  if (my_max_tot) atomicMax(a.max_tot, my_max_tot);
  if (my_bad) *a.bad = 1u;
  if (my_modes) atomicAdd(a.bad + 1, my_modes);
  if (my_max_len > 8u) atomicMax(a.max_tot + 5, my_max_len);
}

// ------------------------------------------------------------------------------------------------
// The signed method's inspector, block-staged like k_stats_ie2 (round 4).  A joined path is two half-rows; every 16-lane
// group works on ONE (path, half) -- a "virtual row" -- so a wave owns 32 consecutive joined paths = 64 virtual rows, the
// group's half is fixed (group & 1: a group that stays on the same paths0 row keeps that half's words), and the per-half
// outputs (carriers, carriers among the cases, list info, list slot) are staged in LDS exactly as the unsigned kernel
// stages its per-path ones.  What needs both halves -- the observed score vt[case_pos][ctrl_neg] + vt[case_neg][ctrl_pos]
// (methods.h:255), the reported counts (:256-257) -- is put together per path when the block is written out.
// Same outputs as k_stats_ie<2, .>, bit for bit (GCRE_STATS_V1=1 runs that one).
// ------------------------------------------------------------------------------------------------
template <int NL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NL <= 3 ? GCRE_STATS_WAVES : 2))) void k_stats_ie2s(const StatsArgs a) {
  typedef u64 __attribute__((ext_vector_type(2))) u64x2;
  constexpr u32 kNoRange = 0xffffffffu;
  constexpr u32 kOverChunk = 2048;
  __shared__ u32 slot_lds[4][64 * 8];   // the 64 virtual rows' list slots
  __shared__ u32 out_lds[4][4][64];     // per virtual row: carriers, carriers among the cases, linfo, lover
  __shared__ u32 pair_lds[4][4][3][32 * NL];
  __shared__ u32 meta_lds[4][4][32];    // per path: paths0 row, reduced row | flip, range, swap of paths1
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sl = lane & 15, grp = lane >> 4;
  const int h = grp & 1;                // the half this group works on, in the joined path's orientation
  const u32 gsh = (u32)grp * 16u, ltm = (1u << sl) - 1u;
  u32 (*pairs)[32 * NL] = pair_lds[wv][grp];
  u32 (*meta)[32] = meta_lds[wv];
  const i64 wave = (i64)blockIdx.x * 4 + wv;
  const i64 nwaves = (i64)gridDim.x * 4;
  const int Wp = a.Wp;
  u32* slots = slot_lds[wv];
  u32 (*outs)[64] = out_lds[wv];
  u32 my_max_tot = 0, my_modes = 0, my_max_len = 0;
  bool my_bad = false;
  u32 chunk_at = 0u, chunk_left = 0u;
  __shared__ u64 cm_lds[32 * NL];
  for (int k = (int)threadIdx.x; k < 32 * NL; k += 256) cm_lds[k] = k < Wp ? a.case_mask[k] : 0;
  __syncthreads();
  const i64 nblocks = (a.count + 31) / 32;
  for (i64 blk = wave; blk < nblocks; blk += nwaves) {
    const i64 base = blk * 32;
    {
      // ---- the 32 paths' row numbers (lanes t and t + 32 both read path base + t) ----
      const i64 iq = base + (lane & 31) < a.count ? base + (lane & 31) : a.count - 1;
      const u32 r0v = a.row0[iq];
      const u32 r1v = a.row1[iq];
      const u32 zraw = a.zindex ? (u32)a.zindex[r1v & 0x7fffffffu] : (r1v & 0x7fffffffu);
      const u32 zflip = (r1v ^ (a.zindex ? zraw : 0u)) & 0x80000000u;
      u32 rngv = kNoRange;   // the uid's row of the excess table, for the first path of a uid in this launch only
      if (a.excess && (iq == 0 || a.row0[iq - 1] != r0v)) rngv = (u32)a.range_of[r0v];
      if (lane < 32) {
        meta[0][lane] = r0v;
        meta[1][lane] = (zraw & 0x7fffffffu) | zflip;
        meta[2][lane] = rngv;
        meta[3][lane] = r1v >> 31;
      }
    }
    __builtin_amdgcn_wave_barrier();
    {
      const u32x4 pad = {a.zoff, a.zoff, a.zoff, a.zoff};
      ((u32x4*)slots)[lane * 2] = pad;
      ((u32x4*)slots)[lane * 2 + 1] = pad;
    }
    u64 xw[NL][2], zw[NL][2];
#pragma unroll
    for (int it = 0; it < NL; it++) xw[it][0] = xw[it][1] = zw[it][0] = zw[it][1] = 0ull;
    u32 n_rzf = 0u, n_rng = kNoRange, n_r0 = 0xffffffffu, n_swap = 0u;
    auto fetch_rows = [&](int itn) {
      const int pln = (itn * 4 + grp) >> 1;
      const u32 r0n = meta[0][pln];
      n_rzf = meta[1][pln];
      n_rng = meta[2][pln];
      n_swap = meta[3][pln];
      const bool new_x = r0n != n_r0;
      n_r0 = r0n;
      const int hz = (n_rzf >> 31) ? 1 - h : h;   // the reduced row's half that lands in half h
      const u64* xn = a.p0 + (size_t)r0n * a.S + (size_t)h * Wp;
      const u64* zn = a.pz + (size_t)(n_rzf & 0x7fffffffu) * a.S + (size_t)hz * Wp;
#pragma unroll
      for (int it = 0; it < NL; it++) {
        const int k = it * 32 + 2 * sl;
        if (k < Wp) {
          if (new_x) {
            const u64x2 v = *(const u64x2*)(xn + k);
            xw[it][0] = v.x; xw[it][1] = v.y;
          }
          const u64x2 v = *(const u64x2*)(zn + k);
          zw[it][0] = v.x; zw[it][1] = v.y;
        }
      }
    };
    fetch_rows(0);
    for (int it8 = 0; it8 < 16; it8++) {
      const int vl = it8 * 4 + grp;                 // the group's virtual row inside the block
      const int pl = vl >> 1;                       // ... and its path
      const bool active = base + pl < a.count;
      if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
      const u32 rng = n_rng, swap = n_swap;
      // the excess row is in paths1's orientation: its half (swap ? 1 - h : h) lands in half h
      const u64* uu = (rng != kNoRange) ? a.excess + (size_t)rng * a.S + (size_t)(swap ? 1 - h : h) * Wp : nullptr;
      u64* out = (a.res && active) ? a.res + (size_t)(a.first + base + pl) * a.S + (size_t)h * Wp : nullptr;
      u32 cc = 0u, dv = 0u;
      u64 stray = 0;
#pragma unroll
      for (int it = 0; it < NL; it++) {
        const int k = it * 32 + 2 * sl;
        u64x2 uv = {0, 0};
        if (k < Wp && uu) uv = *(const u64x2*)(uu + k);
        const u64x2 cmv = *(const u64x2*)(cm_lds + k);
        const u64 cme[2] = {cmv.x, cmv.y};
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const u64 xk = xw[it][e], zk = zw[it][e], uk = e ? uv.y : uv.x;
          stray |= uk & ~xk;
          const u64 j = xk | zk;
          cc += (u32)__popcll(j & cme[e]) | ((u32)__popcll(j) << 16);
          dv += (u32)__popcll(zk & ~xk) | ((u32)__popcll(zk) << 16);
        }
        if (out && k < Wp) *(u64x2*)(out + k) = u64x2{xw[it][0] | zw[it][0], xw[it][1] | zw[it][1]};
      }
      if (stray) my_bad = true;
      const u32 c = row_total(cc, lane), d = row_total(dv, lane);
      const u32 inc = c & 0xffffu, tot = c >> 16;
      const u32 dl = d & 0xffffu, ov = (d >> 16) - dl;
      const u32 mode = a.ie_rule ? ((ov <= 8u || ov < dl) ? 1u : 0u) : ((a.ie_bias >= 0 && ov + (u32)a.ie_bias < dl) ? 1u : 0u);
      const u32 len = mode ? ov : dl;
      const u32 len8 = max(8u, (len + 7u) & ~7u);
      u32 npair = 0u;
#pragma unroll
      for (int it = 0; it < NL; it++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const u64 w = mode ? (zw[it][e] & xw[it][e]) : (zw[it][e] & ~xw[it][e]);
          const bool nz = active && w != 0;
          const u64 bal = __builtin_amdgcn_ballot_w64(nz);
          if (bal == 0ull) continue;
          const u32 m = (u32)(bal >> gsh) & 0xffffu;
          const u32 at = npair + (u32)__builtin_popcount(m & ltm);
          if (nz) {
            pairs[0][at] = (u32)w;
            pairs[1][at] = (u32)(w >> 32);
            pairs[2][at] = (u32)(it * 32 + 2 * sl + e);
          }
          npair += (u32)__builtin_popcount(m);
        }
      }
      fetch_rows(it8 < 15 ? it8 + 1 : 15);   // (changes the NEXT iteration's words and row numbers only)
      if (active && sl == 0) {
        my_modes += mode;
        my_max_tot = max(my_max_tot, tot);
        my_max_len = max(my_max_len, len8);
      }
      const u32 need = (active && len8 > 8u) ? len8 - 8u : 0u;
      const u32 n0 = rdlane(need, 0), n1 = rdlane(need, 16), n2 = rdlane(need, 32), n3 = rdlane(need, 48);
      const u32 nsum = n0 + n1 + n2 + n3;
      u32 ovb = 0u;
      if (nsum != 0u) {
        if (chunk_left < nsum) {
          const u32 grab = nsum > kOverChunk ? nsum : kOverChunk;
          u32 wbase = 0u;
          if (lane == 0) wbase = atomicAdd(a.ov_count, grab);
          chunk_at = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
          chunk_left = grab;
        }
        ovb = chunk_at + (grp > 0 ? n0 : 0u) + (grp > 1 ? n1 : 0u) + (grp > 2 ? n2 : 0u);
        chunk_at += nsum;
        chunk_left -= nsum;
      }
      const bool ov_ok = active && len8 > 8u && (u64)ovb + (len8 - 8u) <= (u64)a.over_cap;
      u32* over = a.over + ovb;
      __builtin_amdgcn_wave_barrier();
      {
        const u32 npmax = max(max(rdlane(npair, 0), rdlane(npair, 16)), max(rdlane(npair, 32), rdlane(npair, 48)));
        u32 cnt = 0u;
        for (u32 q0 = 0u; q0 < npmax; q0 += 16u) {
          const u32 pi = q0 + (u32)sl;
          const bool has = pi < npair;
          u64 w = has ? ((u64)pairs[1][pi] << 32) | (u64)pairs[0][pi] : 0ull;
          const u32 k = has ? pairs[2][pi] : 0u;
          while (__builtin_amdgcn_ballot_w64(w != 0ull) != 0ull) {
            const bool nzb = w != 0ull;
            const u32 m = (u32)(__builtin_amdgcn_ballot_w64(nzb) >> gsh) & 0xffffu;
            const u32 pos = cnt + (u32)__builtin_popcount(m & ltm);
            const u32 b = nzb ? (u32)__builtin_ctzll(w) : 0u;
            w &= w - 1ull;
            const u32 en = (k * 64u + b) << 8;
            if (nzb) {
              if (pos < 8u) slots[vl * 8 + (int)pos] = en;
              else if (ov_ok) over[pos - 8u] = en;
            }
            cnt += (u32)__builtin_popcount(m);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (u32 p = max(len, 8u) + (u32)sl; p < len8; p += 16)   // padding of the overflow part
        if (ov_ok) over[p - 8u] = a.zoff;
      if (active && sl == 0) {
        outs[0][vl] = tot;
        outs[1][vl] = inc;
        outs[2][vl] = len8 | mode | ((len8 - len) << 28);
        outs[3][vl] = ovb;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- per path: both halves together -> observed score, reported counts (lanes 0..31) ----
    if (lane < 32 && base + lane < a.count) {
      const i64 i = base + lane;
      const u32 tot0 = outs[0][2 * lane], tot1 = outs[0][2 * lane + 1];
      // (+) half: case_pos = inc0, ctrl_neg = tot0 - inc0; (-) half: ctrl_pos = inc1, case_neg = tot1 - inc1 (methods.h:182-185)
      const u32 case_pos = outs[1][2 * lane], ctrl_neg = tot0 - case_pos;
      const u32 ctrl_pos = outs[1][2 * lane + 1], case_neg = tot1 - ctrl_pos;
      const double score = a.dvt[(size_t)sp_diag_offset(tot0) + case_pos] + a.dvt[(size_t)sp_diag_offset(tot1) + case_neg];
      a.tot[2 * i] = tot0;
      a.tot[2 * i + 1] = tot1;
      a.cases[i] = case_pos + case_neg;        // methods.h:256-257
      a.ctrls[i] = ctrl_pos + ctrl_neg;
      a.rowz[i] = meta[1][lane];
      a.key[i] = ie_score_key(score);
    }
    // ---- per virtual row: list info and slot, one coalesced store per array ----
    if (base * 2 + lane < a.count * 2) {
      const i64 dd = base * 2 + lane;
      a.linfo[dd] = outs[2][lane];
      a.lover[dd] = outs[3][lane];
      u32x4* dst = (u32x4*)(a.slot + (u64)dd * 8u);
      dst[0] = ((const u32x4*)slots)[lane * 2];
      dst[1] = ((const u32x4*)slots)[lane * 2 + 1];
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (my_max_tot) atomicMax(a.max_tot, my_max_tot);
  if (my_bad) *a.bad = 1u;
  if (my_modes) atomicAdd(a.bad + 1, my_modes);
  if (my_max_len > 8u) atomicMax(a.max_tot + 5, my_max_len);
}

// ------------------------------------------------------------------------------------------------
// The signed method's inspector when every reduced row has an EMPTY half (genes: a gene's carriers sit in one half, the
// other half of its row is zero -- every join below level 5).  Then only one half of a joined path differs from its paths0
// row: k_stats_ie2s spends a round of sixteen lanes on each half of every path, this kernel one round per PATH --
// the half that changes (side of the reduced row, flipped when the relation's sign says so) -- and takes the other half's
// carriers from the paths0 row, counted once per uid when a group of lanes moves on to it (both halves of that row stay in
// registers: consecutive paths of a uid change either half).  A wave owns 64 consecutive paths.  Same outputs as
// k_stats_ie2s except the (unused) overflow offset of an empty list.  a.lz_off: the CSR offsets of the reduced rows' bit
// lists (two lists per row): list 2 r + 1 empty <=> the (-) half of row r is.
// ------------------------------------------------------------------------------------------------
#ifndef GCRE_STATS2H_WAVES
#define GCRE_STATS2H_WAVES 4   // 128 VGPRs, 5 spilled: 7.0 ms per pass on configs[2] geometry against 7.7 at three waves (134, none)
#endif
template <int NL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NL <= 3 ? GCRE_STATS2H_WAVES : 2))) void k_stats_ie2h(const StatsArgs a) {
  typedef u64 __attribute__((ext_vector_type(2))) u64x2;
  constexpr u32 kNoRange = 0xffffffffu;
  constexpr u32 kOverChunk = 2048;
  __shared__ u32 slot_lds[4][64 * 8];   // the 64 paths' list slots (of the half that changes)
  __shared__ u32 out_lds[4][7][64];     // per path: carriers (+), (-), among the cases (+), (-), linfo, lover, the half that changes
  __shared__ u32 pair_lds[4][4][3][32 * NL];
  __shared__ u32 meta_lds[4][5][64];    // per path: paths0 row, reduced row | flip, range, swap of paths1, side of the reduced row
  __shared__ u64 cm_lds[32 * NL];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sl = lane & 15, grp = lane >> 4;
  const u32 gsh = (u32)grp * 16u, ltm = (1u << sl) - 1u;
  u32 (*pairs)[32 * NL] = pair_lds[wv][grp];
  u32 (*meta)[64] = meta_lds[wv];
  const i64 wave = (i64)blockIdx.x * 4 + wv;
  const i64 nwaves = (i64)gridDim.x * 4;
  const int Wp = a.Wp;
  u32* slots = slot_lds[wv];
  u32 (*outs)[64] = out_lds[wv];
  u32 my_max_tot = 0, my_modes = 0, my_max_len = 0;
  bool my_bad = false;
  u32 chunk_at = 0u, chunk_left = 0u;
  for (int k = (int)threadIdx.x; k < 32 * NL; k += 256) cm_lds[k] = k < Wp ? a.case_mask[k] : 0;
  __syncthreads();
  // an empty list as the inspector writes it: no entries, eight of padding (the bias rule of the signed method: delta list)
  const u32 empty_mode = a.ie_rule ? 1u : 0u;   // (overlap 0, delta 0: the slot rule says overlap list, the bias rule delta list)
  const u32 empty_linfo = 8u | empty_mode | (8u << 28);
  const i64 nblocks = (a.count + 63) / 64;
  for (i64 blk = wave; blk < nblocks; blk += nwaves) {
    const i64 base = blk * 64;
    {
      // ---- the 64 paths' row numbers, one coalesced load per array (lane t <-> path base + t) ----
      const i64 iq = base + lane < a.count ? base + lane : a.count - 1;
      const u32 r0v = a.row0[iq];
      const u32 r1v = a.row1[iq];
      const u32 zraw = a.zindex ? (u32)a.zindex[r1v & 0x7fffffffu] : (r1v & 0x7fffffffu);
      const u32 zflip = (r1v ^ (a.zindex ? zraw : 0u)) & 0x80000000u;
      const u32 zrow = zraw & 0x7fffffffu;
      u32 rngv = kNoRange;   // the uid's row of the excess table, for the first path of a uid in this launch only
      if (a.excess && (iq == 0 || a.row0[iq - 1] != r0v)) rngv = (u32)a.range_of[r0v];
      const u32 side = a.lz_off[2 * (size_t)zrow + 2] > a.lz_off[2 * (size_t)zrow + 1] ? 1u : 0u;   // the row's (-) half has carriers
      meta[0][lane] = r0v;
      meta[1][lane] = zrow | zflip;
      meta[2][lane] = rngv;
      meta[3][lane] = r1v >> 31;
      meta[4][lane] = side;
    }
    __builtin_amdgcn_wave_barrier();
    {
      const u32x4 pad = {a.zoff, a.zoff, a.zoff, a.zoff};
      ((u32x4*)slots)[lane * 2] = pad;
      ((u32x4*)slots)[lane * 2 + 1] = pad;
    }
    u64 xw[2][NL][2], zw[NL][2];
#pragma unroll
    for (int it = 0; it < NL; it++) xw[0][it][0] = xw[0][it][1] = xw[1][it][0] = xw[1][it][1] = zw[it][0] = zw[it][1] = 0ull;
    u32 n_rzf = 0u, n_rng = kNoRange, n_r0 = 0xffffffffu, n_swap = 0u, n_hc = 0u;
    bool n_newx = false;
    auto fetch_rows = [&](int itn) {
      const int pln = itn * 4 + grp;
      const u32 r0n = meta[0][pln];
      n_rzf = meta[1][pln];
      n_rng = meta[2][pln];
      n_swap = meta[3][pln];
      const u32 side = meta[4][pln];
      n_hc = (n_rzf >> 31) ? 1u - side : side;   // the half of the joined path the reduced row's carriers land in
      n_newx = r0n != n_r0;
      n_r0 = r0n;
      const u64* xn = a.p0 + (size_t)r0n * a.S;
      const u64* zn = a.pz + (size_t)(n_rzf & 0x7fffffffu) * a.S + (size_t)side * Wp;
#pragma unroll
      for (int it = 0; it < NL; it++) {
        const int k = it * 32 + 2 * sl;
        if (k < Wp) {
          if (n_newx) {
            const u64x2 v0 = *(const u64x2*)(xn + k);
            const u64x2 v1 = *(const u64x2*)(xn + Wp + k);
            xw[0][it][0] = v0.x; xw[0][it][1] = v0.y;
            xw[1][it][0] = v1.x; xw[1][it][1] = v1.y;
          }
          const u64x2 v = *(const u64x2*)(zn + k);
          zw[it][0] = v.x; zw[it][1] = v.y;
        }
      }
    };
    fetch_rows(0);
    u32 tx0 = 0u, tx1 = 0u;   // carriers | carriers among the cases << 16 of the paths0 row's halves
    for (int it4 = 0; it4 < 16; it4++) {
      const int pl = it4 * 4 + grp;                 // the group's path inside the block
      const bool active = base + pl < a.count;
      if (__builtin_amdgcn_ballot_w64(active) == 0ull) break;
      const u32 rng = n_rng, swap = n_swap, hc = n_hc;
      const bool newx = n_newx;
      // ---- a new paths0 row: the carriers of both its halves; (first path of a uid) the check of the hint, both halves ----
      if (__builtin_amdgcn_ballot_w64(newx) != 0ull) {
        u32 c0 = 0u, c1 = 0u;
        u64 stray = 0;
        // the excess row is in paths1's orientation: its half (swap ? 1 - h : h) lands in half h
        const u64* uu = (newx && rng != kNoRange) ? a.excess + (size_t)rng * a.S : nullptr;
#pragma unroll
        for (int it = 0; it < NL; it++) {
          const int k = it * 32 + 2 * sl;
          const u64x2 cmv = *(const u64x2*)(cm_lds + k);
          c0 += ((u32)__popcll(xw[0][it][0]) + (u32)__popcll(xw[0][it][1])) |
                (((u32)__popcll(xw[0][it][0] & cmv.x) + (u32)__popcll(xw[0][it][1] & cmv.y)) << 16);
          c1 += ((u32)__popcll(xw[1][it][0]) + (u32)__popcll(xw[1][it][1])) |
                (((u32)__popcll(xw[1][it][0] & cmv.x) + (u32)__popcll(xw[1][it][1] & cmv.y)) << 16);
          if (uu && k < Wp) {
            const u64x2 u0 = *(const u64x2*)(uu + (size_t)(swap ? 1 : 0) * Wp + k);   // lands in half 0
            const u64x2 u1 = *(const u64x2*)(uu + (size_t)(swap ? 0 : 1) * Wp + k);   // lands in half 1
            stray |= (u0.x & ~xw[0][it][0]) | (u0.y & ~xw[0][it][1]) | (u1.x & ~xw[1][it][0]) | (u1.y & ~xw[1][it][1]);
          }
        }
        if (stray) my_bad = true;
        const u32 t0 = row_total(c0, lane), t1 = row_total(c1, lane);
        if (newx) { tx0 = t0; tx1 = t1; }
      }
      u64* out = (a.res && active) ? a.res + (size_t)(a.first + base + pl) * a.S : nullptr;
      // ---- the half that changes ----
      u32 cc = 0u, dv = 0u;
      u64 xk_[NL][2];
#pragma unroll
      for (int it = 0; it < NL; it++) {
        const int k = it * 32 + 2 * sl;
        const u64x2 cmv = *(const u64x2*)(cm_lds + k);
        const u64 cme[2] = {cmv.x, cmv.y};
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const u64 xk = hc ? xw[1][it][e] : xw[0][it][e], zk = zw[it][e];
          xk_[it][e] = xk;
          const u64 j = xk | zk;
          cc += (u32)__popcll(j & cme[e]) | ((u32)__popcll(j) << 16);
          dv += (u32)__popcll(zk & ~xk) | ((u32)__popcll(zk) << 16);
        }
        if (out && k < Wp) {   // the kept row: the changed half joined, the other as it was
          const u64x2 jc = u64x2{xk_[it][0] | zw[it][0], xk_[it][1] | zw[it][1]};
          const u64x2 ju = hc ? u64x2{xw[0][it][0], xw[0][it][1]} : u64x2{xw[1][it][0], xw[1][it][1]};
          *(u64x2*)(out + (size_t)hc * Wp + k) = jc;
          *(u64x2*)(out + (size_t)(1u - hc) * Wp + k) = ju;
        }
      }
      const u32 c = row_total(cc, lane), d = row_total(dv, lane);
      const u32 inc = c & 0xffffu, tot = c >> 16;
      const u32 dl = d & 0xffffu, ov = (d >> 16) - dl;
      const u32 mode = a.ie_rule ? ((ov <= 8u || ov < dl) ? 1u : 0u) : ((a.ie_bias >= 0 && ov + (u32)a.ie_bias < dl) ? 1u : 0u);
      const u32 len = mode ? ov : dl;
      const u32 len8 = max(8u, (len + 7u) & ~7u);
      u32 npair = 0u;
#pragma unroll
      for (int it = 0; it < NL; it++) {
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const u64 w = mode ? (zw[it][e] & xk_[it][e]) : (zw[it][e] & ~xk_[it][e]);
          const bool nz = active && w != 0;
          const u64 bal = __builtin_amdgcn_ballot_w64(nz);
          if (bal == 0ull) continue;
          const u32 m = (u32)(bal >> gsh) & 0xffffu;
          const u32 at = npair + (u32)__builtin_popcount(m & ltm);
          if (nz) {
            pairs[0][at] = (u32)w;
            pairs[1][at] = (u32)(w >> 32);
            pairs[2][at] = (u32)(it * 32 + 2 * sl + e);
          }
          npair += (u32)__builtin_popcount(m);
        }
      }
      // the other half's carriers: the paths0 row's
      const u32 tu = hc ? tx0 : tx1;
      const u32 tot_u = tu & 0xffffu, inc_u = tu >> 16;
      fetch_rows(it4 < 15 ? it4 + 1 : 15);   // (changes the NEXT iteration's words and row numbers only)
      if (active && sl == 0) {
        my_modes += mode + empty_mode;
        my_max_tot = max(my_max_tot, max(tot, tot_u));
        my_max_len = max(my_max_len, len8);
      }
      const u32 need = (active && len8 > 8u) ? len8 - 8u : 0u;
      const u32 n0 = rdlane(need, 0), n1 = rdlane(need, 16), n2 = rdlane(need, 32), n3 = rdlane(need, 48);
      const u32 nsum = n0 + n1 + n2 + n3;
      u32 ovb = 0u;
      if (nsum != 0u) {
        if (chunk_left < nsum) {
          const u32 grab = nsum > kOverChunk ? nsum : kOverChunk;
          u32 wbase = 0u;
          if (lane == 0) wbase = atomicAdd(a.ov_count, grab);
          chunk_at = (u32)__builtin_amdgcn_readfirstlane((int)wbase);
          chunk_left = grab;
        }
        ovb = chunk_at + (grp > 0 ? n0 : 0u) + (grp > 1 ? n1 : 0u) + (grp > 2 ? n2 : 0u);
        chunk_at += nsum;
        chunk_left -= nsum;
      }
      const bool ov_ok = active && len8 > 8u && (u64)ovb + (len8 - 8u) <= (u64)a.over_cap;
      u32* over = a.over + ovb;
      __builtin_amdgcn_wave_barrier();
      {
        const u32 npmax = max(max(rdlane(npair, 0), rdlane(npair, 16)), max(rdlane(npair, 32), rdlane(npair, 48)));
        u32 cnt = 0u;
        for (u32 q0 = 0u; q0 < npmax; q0 += 16u) {
          const u32 pi = q0 + (u32)sl;
          const bool has = pi < npair;
          u64 w = has ? ((u64)pairs[1][pi] << 32) | (u64)pairs[0][pi] : 0ull;
          const u32 k = has ? pairs[2][pi] : 0u;
          while (__builtin_amdgcn_ballot_w64(w != 0ull) != 0ull) {
            const bool nzb = w != 0ull;
            const u32 m = (u32)(__builtin_amdgcn_ballot_w64(nzb) >> gsh) & 0xffffu;
            const u32 pos = cnt + (u32)__builtin_popcount(m & ltm);
            const u32 b = nzb ? (u32)__builtin_ctzll(w) : 0u;
            w &= w - 1ull;
            const u32 en = (k * 64u + b) << 8;
            if (nzb) {
              if (pos < 8u) slots[pl * 8 + (int)pos] = en;
              else if (ov_ok) over[pos - 8u] = en;
            }
            cnt += (u32)__builtin_popcount(m);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (u32 p = max(len, 8u) + (u32)sl; p < len8; p += 16)   // padding of the overflow part
        if (ov_ok) over[p - 8u] = a.zoff;
      if (active && sl == 0) {
        outs[0][pl] = hc ? tot_u : tot;
        outs[1][pl] = hc ? tot : tot_u;
        outs[2][pl] = hc ? inc_u : inc;
        outs[3][pl] = hc ? inc : inc_u;
        outs[4][pl] = len8 | mode | ((len8 - len) << 28);
        outs[5][pl] = ovb;
        outs[6][pl] = hc;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- per path: both halves together -> observed score, reported counts ----
    if (base + lane < a.count) {
      const i64 i = base + lane;
      const u32 tot0 = outs[0][lane], tot1 = outs[1][lane];
      // (+) half: case_pos = inc0, ctrl_neg = tot0 - inc0; (-) half: ctrl_pos = inc1, case_neg = tot1 - inc1 (methods.h:182-185)
      const u32 case_pos = outs[2][lane], ctrl_neg = tot0 - case_pos;
      const u32 ctrl_pos = outs[3][lane], case_neg = tot1 - ctrl_pos;
      const double score = a.dvt[(size_t)sp_diag_offset(tot0) + case_pos] + a.dvt[(size_t)sp_diag_offset(tot1) + case_neg];
      a.tot[2 * i] = tot0;
      a.tot[2 * i + 1] = tot1;
      a.cases[i] = case_pos + case_neg;        // methods.h:256-257
      a.ctrls[i] = ctrl_pos + ctrl_neg;
      a.rowz[i] = meta[1][lane];
      a.key[i] = ie_score_key(score);
    }
    // ---- per virtual row (two per path): list info and slot, coalesced; the half that did not change has the empty list ----
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int v = j * 64 + lane, pv = v >> 1;
      if (base + pv < a.count) {
        const i64 dd = base * 2 + v;
        const bool ch = outs[6][pv] == (u32)(v & 1);
        a.linfo[dd] = ch ? outs[4][pv] : empty_linfo;
        a.lover[dd] = ch ? outs[5][pv] : 0u;
        const u32x4 pad = {a.zoff, a.zoff, a.zoff, a.zoff};
        u32x4* dst = (u32x4*)(a.slot + (u64)dd * 8u);
        dst[0] = ch ? ((const u32x4*)slots)[pv * 2] : pad;
        dst[1] = ch ? ((const u32x4*)slots)[pv * 2 + 1] : pad;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (my_max_tot) atomicMax(a.max_tot, my_max_tot);
  if (my_bad) *a.bad = 1u;
  if (my_modes) atomicAdd(a.bad + 1, my_modes);
  if (my_max_len > 8u) atomicMax(a.max_tot + 5, my_max_len);
}

hipError_t launch_stats_ie(const StatsArgs& a, int method, hipStream_t stream) {
  if (a.count == 0) return hipSuccess;
  static const bool v1 = std::getenv("GCRE_STATS_V1") != nullptr;   // the per-path form (cross-check)
  if (method == 1 && !v1 && a.Wp <= 160) {
    const i64 nb = (a.count + 63) / 64;
    const i64 blocks = (nb + 3) / 4;
    const dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), block(256);
    const int nl = (a.Wp + 31) / 32;
    if (nl <= 1) hipLaunchKernelGGL((k_stats_ie2<1>), grid, block, 0, stream, a);
    else if (nl == 2) hipLaunchKernelGGL((k_stats_ie2<2>), grid, block, 0, stream, a);
    else if (nl == 3) hipLaunchKernelGGL((k_stats_ie2<3>), grid, block, 0, stream, a);
    else if (nl == 4) hipLaunchKernelGGL((k_stats_ie2<4>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_stats_ie2<5>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  // the signed method with one-sided reduced rows: one round per path (k_stats_ie2h); GCRE_STATS_BOTH_HALVES=1: k_stats_ie2s
  static const bool both_halves = std::getenv("GCRE_STATS_BOTH_HALVES") != nullptr;
  if (method == 2 && !v1 && !both_halves && a.Wp <= 160 && a.lz_off) {
    const i64 nb = (a.count + 63) / 64;
    const i64 blocks = (nb + 3) / 4;
    const dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), block(256);
    const int nl = (a.Wp + 31) / 32;
    if (nl <= 1) hipLaunchKernelGGL((k_stats_ie2h<1>), grid, block, 0, stream, a);
    else if (nl == 2) hipLaunchKernelGGL((k_stats_ie2h<2>), grid, block, 0, stream, a);
    else if (nl == 3) hipLaunchKernelGGL((k_stats_ie2h<3>), grid, block, 0, stream, a);
    else if (nl == 4) hipLaunchKernelGGL((k_stats_ie2h<4>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_stats_ie2h<5>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  if (method == 2 && !v1 && a.Wp <= 160) {
    const i64 nb = (a.count + 31) / 32;
    const i64 blocks = (nb + 3) / 4;
    const dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), block(256);
    const int nl = (a.Wp + 31) / 32;
    if (nl <= 1) hipLaunchKernelGGL((k_stats_ie2s<1>), grid, block, 0, stream, a);
    else if (nl == 2) hipLaunchKernelGGL((k_stats_ie2s<2>), grid, block, 0, stream, a);
    else if (nl == 3) hipLaunchKernelGGL((k_stats_ie2s<3>), grid, block, 0, stream, a);
    else if (nl == 4) hipLaunchKernelGGL((k_stats_ie2s<4>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_stats_ie2s<5>), grid, block, 0, stream, a);
    return hipGetLastError();
  }
  const i64 blocks = (a.count + 15) / 16;   // 4 waves x 4 paths per block and pass
  const dim3 grid((unsigned)(blocks < 256 * 16 ? blocks : 256 * 16)), block(256);
  const int nit = (a.Wp + 15) / 16;
#define GCRE_ST(MM, NN) hipLaunchKernelGGL((k_stats_ie<MM, NN>), grid, block, 0, stream, a)
  if (method == 1) {
    if (nit <= 1) GCRE_ST(1, 1); else if (nit <= 2) GCRE_ST(1, 2); else if (nit <= 3) GCRE_ST(1, 3); else if (nit <= 4) GCRE_ST(1, 4);
    else if (nit <= 5) GCRE_ST(1, 5); else if (nit <= 6) GCRE_ST(1, 6); else if (nit <= 8) GCRE_ST(1, 8); else GCRE_ST(1, 0);
  } else {
    if (nit <= 1) GCRE_ST(2, 1); else if (nit <= 2) GCRE_ST(2, 2); else if (nit <= 3) GCRE_ST(2, 3); else if (nit <= 4) GCRE_ST(2, 4);
    else if (nit <= 5) GCRE_ST(2, 5); else GCRE_ST(2, 0);
  }
#undef GCRE_ST
  return hipGetLastError();
}

}  // namespace gcre
