// gcre_ie2.hip -- the signed method's pruned permutation (null) kernel on count planes (gfx950).
//
// Reference: JoinMethod2::score_permute, src/methods.h:130-232.  A joined path has a (+) and a (-) half with their own
// counts a, b and carrier totals tp, tn; its null score is (float)(F + G), F = vtmax[a][tp - a], G = vtmax[tn - b][b]
// (methods.h:220-230) -- two cells of two different table diagonals, summed in f64, rounded once.
//
// Counts come from the planes as in gcre_ie.hip:  count_h = N0[idx][h] + Nz[z][h'] - popc(p0_h & z_h' & mask).
//
// What this kernel is organised around (round 4):
//
// * ONE half changes.  Every join of the ProcessPaths sequence up to level 4 adds ONE gene, and a gene sits in one half:
//   the other half of the joined path IS the paths0 half -- same counts, same carrier total, for every such path of the
//   segment (the inspector marks it with an empty delta list).  A segment's paths are therefore sorted into three classes
//   with two ballots -- (+) half changes, (-) half changes, both change (level 5 adds a 2-gene path; a join without a
//   reduced operand) -- and each class is walked by a scalar loop over its lanes.  For the one-sided classes everything
//   that concerns the unchanged half (its interval tests, its diagonal) is done once per class, the base counters of the
//   two halves trade places (v_swap) so that one loop body serves both orientations, and a path costs one half's rows,
//   planes and adds.  Round 3's kernel walked both halves of every path: ~390 instructions per path-tile, 119 spilled
//   scalars.
// * The cover of  F + G <= theta  is a STAIRCASE of NS steps (GCRE_M2_STEPS, 3): levels r_0 < .. < r_{NS-1} with
//   r_s + r_{NS-1-s} <= theta; a permutation with F <= r_s and G <= r_{NS-1-s} for some s cannot raise its maximum (the
//   f64 sum is <= theta, rounding to f32 is monotone).  Round 3 used two steps (theta/3, 2 theta/3); a third costs one
//   more interval test on the changed half and leaves 4 exp(-3 theta / 4) instead of 3 exp(-2 theta / 3) of the
//   permutations to look up.  What fails every step is looked up exactly: both cells gathered in f64, added, rounded,
//   clamped at 0 -- through the per-wave look-up queue of round 3.
// * Kept rows leave a RECIPE here too (REC): row r = row rec_row0[r] of set A | row rec_rowz[r] of set Z, per half the
//   producing join's list.  The segment's base counters are rebuilt from the planes of A and Z in the prologue; the kept
//   set's own planes (68 GB per pass at configs[2] geometry) are neither written nor read.
#include "gcre_ie_common.h"

#ifndef GCRE_M2_STEPS
#define GCRE_M2_STEPS 3
#endif

namespace gcre {

template <int L, int GZ, bool OUT, bool REC>
__global__ __launch_bounds__(64 * kIeWaves) __attribute__((amdgpu_waves_per_eu(L <= 10 ? 4 : 3))) void k_null_ie_m2(const IeArgs a) {
  constexpr int NS = GCRE_M2_STEPS;
  static_assert(NS == 2 || NS == 3, "two or three steps");
  constexpr int LP = (L + 3) / 4 * 4;
  constexpr int LZ = 4 * GZ;   // planes of an added row (GZ groups; the launch picks GZ >= a.gz)
  static_assert(LZ <= LP, "added rows have no more planes than joined paths");
  static_assert(L >= 8 && L <= 16, "8 to 16 counter planes");
  typedef u32 __attribute__((ext_vector_type(8))) u32x8;
  __shared__ u32 nmax_lds[kIeWaves][32 * 64];
  // Look-ups are QUEUED, not made where they are found (round 3): a permutation that fails every step only leaves (cell
  // of F, cell of G, slot of its maximum) in the wave's LDS queue; when 64 are waiting (or the tile ends) every lane takes
  // one, and ONE round trip serves 64 look-ups.  Maxima may lag a queue behind: a threshold read meanwhile is only lower.
  constexpr u32 kLqCap = 128u;
  __shared__ u32 lq_lds[kIeWaves][3][kLqCap];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane4 = (u32)lane * 4u;
  u32* nm = nmax_lds[wave] + lane;
#pragma unroll
  for (int q = 0; q < 32; q++) nm[q * 64] = 0u;
  const SparseSeg GCRE_CONSTANT* segs = (const SparseSeg GCRE_CONSTANT*)a.segs;
  u32 (*lq)[kLqCap] = lq_lds[wave];
  u32 lq_n = 0u;   // entries waiting (wave-uniform)
  bool dirty = false;
  auto lq_drain = [&]() {
    for (u32 base = 0u; base < lq_n; base += 64u) {
      const u32 i = base + (u32)lane;
      if (i < lq_n) {
        const double f64 = a.d64[lq[0][i]] + a.d64[lq[1][i]];   // vtmax[a][tp - a] + vtmax[tn - b][b], methods.h:227
        float f = (float)f64;
        f = (f > 0.0f) ? f : 0.0f;
        __hip_atomic_fetch_max(nmax_lds[wave] + lq[2][i], __float_as_uint(f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      }
    }
    if (lq_n != 0u) dirty = true;
    lq_n = 0u;
  };
  // the base of the added rows' planes in scalar registers: their loads take the scalar-base buffer form
  const char* k_planesz = (const char*)(((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)((u64)a.planesz >> 32)) << 32) |
                                        (u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(u64)a.planesz));

  int cur_kt = -1;
  u32 valid = 0u;
  // the staircase: ladder rows r_0 <= .. <= r_{NS-1} with r_s + r_{NS-1-s} <= 2 j, rows in units of 1 / (2 kLadderPerUnit)
  u32 lad[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) lad[s] = (a.lad_mode == 0) ? 0u : (u32)(kLadder2Levels - 1 + a.lad_mode) * (u32)a.ladder_stride;
  const u32 lad_keep = (u32)kLadder2Levels * (u32)a.ladder_stride;
  u32 n_slow = 0u;
  __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)a.mt, 0, 0x7fffffff, 0x00020000);

  auto exchange = [&]() {
    lq_drain();   // what the queue still holds belongs to the maxima that go out
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
    int j = (int)(__uint_as_float(theta) * (float)kLadderPerUnit);   // theta >= j / kLadderPerUnit = 2 j rows
    j = j < 0 ? 0 : (j > kLadderLevels - 1 ? kLadderLevels - 1 : j);
    int r[NS];
    if constexpr (NS == 2) {
      r[0] = (2 * j) / 3;
      r[1] = 2 * j - r[0];
    } else {
      r[0] = j / 2;
      r[1] = j;
      r[2] = 2 * j - r[0];
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
      const int rs = r[s] > kLadder2Levels - 1 ? kLadder2Levels - 1 : r[s];   // a lower row is a lower threshold: still exact
      lad[s] = (u32)rs * (u32)a.ladder_stride;
    }
  };
  auto flush_tile = [&]() {
    lq_drain();
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
  auto load_groups = [&](u32 (&P)[LP], const u32* planes, u64 unit, int groups) {
    const u32x4* src = (const u32x4*)(planes + unit * 256u) + lane;
#pragma unroll
    for (int j = 0; j < LP / 4; j++) {
      u32x4 v = {0u, 0u, 0u, 0u};
      if (j < groups) v = src[j * 64];
      P[4 * j + 0] = v.x; P[4 * j + 1] = v.y; P[4 * j + 2] = v.z; P[4 * j + 3] = v.w;
    }
  };
  // permutations whose count lies outside [lo, hi] (bounds as scalar masks, two borrow chains)
  auto outside = [&](const auto& C, u32 lh) -> u32 {   // C: at least L planes
    const u32 lo = lh & 0xffffu, hi = lh >> 16;
    u32 blo = 0u, bhi = 0u;
#pragma unroll
    for (int l = 0; l < L; l++) {
      const u32 kl = (u32)__builtin_amdgcn_sbfe((int)lo, l, 1);
      const u32 kh = (u32)__builtin_amdgcn_sbfe((int)hi, l, 1);
      blo = borrow3(C[l], kl, blo);
      bhi = borrow3(kh, C[l], bhi);
    }
    return blo | bhi;
  };
  // S = how many of a list's mask rows a permutation has set: the 8 rows in y, then (long lists) further blocks of 8
  auto list_sum = [&](u32 (&S)[L], const u32 (&y)[8], u32 len, const u32 GCRE_CONSTANT* more) {
    u32 S4[4];
    sum8(y, S4);
#pragma unroll
    for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
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
  };
  // C = Bx + Nz - S (overlap list) or Bx + S (delta list), info = the list's linfo word
  auto add_half = [&](u32 (&C)[L], const auto& Bx, const u32 (&y)[8], const u32 (&Z)[LZ], u32 info, const u32 GCRE_CONSTANT* more) {
    const u32 len = info & kLinfoLenMask;
    const bool overlap = (info & 1u) != 0u;
    if (len <= 8u) {
      u32 S4[4];
      sum8(y, S4);
      if (overlap) {   // Nz - S >= 0: the overlap is part of the added row
        u32 T[LZ];
        u32 bw = 0u;
#pragma unroll
        for (int l = 0; l < LZ; l++) {
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
          if (l < LZ) {
            C[l] = xor3(Bx[l], T[l < LZ ? l : 0], cy);
            cy = majority(Bx[l], T[l < LZ ? l : 0], cy);
          } else {
            C[l] = Bx[l] ^ cy;
            cy = Bx[l] & cy;
          }
        }
      } else {
        u32 cy = 0u;
#pragma unroll
        for (int l = 0; l < L; l++) {
          if (l < 4) {
            C[l] = xor3(Bx[l], S4[l < 4 ? l : 0], cy);
            cy = majority(Bx[l], S4[l < 4 ? l : 0], cy);
          } else {
            C[l] = Bx[l] ^ cy;
            cy = Bx[l] & cy;
          }
        }
      }
      return;
    }
    u32 S[L];
    list_sum(S, y, len, more);
    if (overlap) {
      u32 cy = 0u, bw = 0u;
#pragma unroll
      for (int l = 0; l < L; l++) {
        const u32 zl = (l < LZ) ? Z[l < LZ ? l : 0] : 0u;
        const u32 s1_ = xor3(Bx[l], zl, cy);
        cy = majority(Bx[l], zl, cy);
        C[l] = xor3(s1_, S[l], bw);
        bw = borrow3(s1_, S[l], bw);
      }
    } else {
      u32 cy = 0u;
#pragma unroll
      for (int l = 0; l < L; l++) {
        C[l] = xor3(Bx[l], S[l], cy);
        cy = majority(Bx[l], S[l], cy);
      }
    }
  };
  // the loads of one list: its first 8 mask rows -- only the entries that are not padding, in steps of four (a wave-load
  // costs the memory pipe the same whatever it fetches) -- and, for an overlap list, the planes of the added row-half
  auto issue = [&](u32 info, u32 zu, const u32x8 offs, u32 (&yy)[8], u32 (&ZZ)[LZ]) {
    const u32 real = (info & kLinfoLenMask) - (info >> 28);
    if (real > 4u) {
#pragma unroll
      for (int j = 0; j < 8; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
    } else if (real > 0u) {
#pragma unroll
      for (int j = 0; j < 4; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
#pragma unroll
      for (int j = 4; j < 8; j++) yy[j] = 0u;
    } else {
#pragma unroll
      for (int j = 0; j < 8; j++) yy[j] = 0u;
    }
    if (info & 1u) {
      __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(k_planesz + (u64)zu * 1024u), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < GZ; j++) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (j < a.gz) v = __builtin_amdgcn_raw_buffer_load_b128(rz, lane4 * 4u + (u32)j * 1024u, 0, 0);
        ZZ[4 * j + 0] = v.x; ZZ[4 * j + 1] = v.y; ZZ[4 * j + 2] = v.z; ZZ[4 * j + 3] = v.w;
      }
    }
  };
  // the permutations in m leave their two cells and their slot in the look-up queue, one per lane and round
  auto enqueue = [&](u32 m, const auto& Ca, const auto& Cb, u32 da, u32 db) {
    while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
      if (lq_n + 64u > kLqCap) lq_drain();
      const bool has = m != 0u;
      const u32 bb = has ? (u32)__builtin_ctz(m) : 0u;
      m &= m - 1u;
      u32 ca = 0u, cb = 0u;
#pragma unroll
      for (int l = 0; l < L; l++) {
        ca |= ((Ca[l] >> bb) & 1u) << l;
        cb |= ((Cb[l] >> bb) & 1u) << l;
      }
      const u64 hm = __builtin_amdgcn_ballot_w64(has);
      const u32 pos = lq_n + __builtin_amdgcn_mbcnt_hi((u32)(hm >> 32), __builtin_amdgcn_mbcnt_lo((u32)hm, 0u));
      if (has) {
        lq[0][pos] = da + ca;
        lq[1][pos] = db + cb;
        lq[2][pos] = bb * 64u + (u32)lane;
      }
      lq_n += (u32)__builtin_popcountll(hm);
    }
  };

  // work queues as in k_null_ie_m1
  __shared__ u32 wq_state[kIeWaves][8];
  WorkQueue wq;
  wq.st = wq_state[wave];
  wq.init(a.queue, (u32)((a.seg_end - a.seg_begin + a.batch - 1) / a.batch), (u32)a.nkt);
  wq.select(blockIdx.x & 7u);   // workgroups are dealt round the XCDs: blocks b and b + 8 share an L2
  u32 ticket = wq.take(lane);
  int since = 0, period = 1;
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
      if (a.lad_mode == 0) {
#pragma unroll
        for (int s = 0; s < NS; s++) lad[s] = 0u;
      }
      since = 0;
      period = 1;
    }
    for (i64 sidx = s_lo; sidx < s_hi; sidx++) {
      const u32 row0 = segs[sidx].row0;
      const u32 first = segs[sidx].first;
      const u32 npaths = segs[sidx].n;
      if (a.lad_mode == 0 && ++since >= period) {
        exchange();
        since = 0;
        period = period < kIeRefresh ? period * 2 : kIeRefresh;
      }
      // ---- per-path metadata of the segment: lane t <-> joined path first + t.  Index 0 / 1 of every pair below is the
      // (+) / (-) half until the one-sided pass of the (-) half swaps them ----
      const bool inl = (u32)lane < npaths;
      const u32 qv = first + (inl ? (u32)lane : 0u);
      const u32 rzv = a.rowz[qv];
      const bool mine = (u64)sidx < (u64)a.score_segs;   // else: rows of another shard, the ladder's all-inside row
      u32 infov[2], lovv[2], zunit[2], totv[2], ladv[NS][2];
#pragma unroll
      for (int h = 0; h < 2; h++) {
        infov[h] = a.linfo[(u64)qv * 2 + h];
        lovv[h] = a.lover[(u64)qv * 2 + h];
        const u32 hz = (rzv >> 31) ? (u32)(1 - h) : (u32)h;
        zunit[h] = ((u32)kt * (u32)a.rowsz + (rzv & 0x7fffffffu) * 2u + hz) * (u32)a.gz;
        totv[h] = a.tot[(u64)qv * 2 + h];
#pragma unroll
        for (int s = 0; s < NS; s++) ladv[s][h] = a.ladder[(mine ? lad[s] : lad_keep) + totv[h]];
      }
      // classes: a half with an empty delta list is the paths0 half (the added row has nothing there, or nothing new)
      const bool e0 = !(infov[0] & 1u) && (infov[0] & kLinfoLenMask) == (infov[0] >> 28);
      const bool e1 = !(infov[1] & 1u) && (infov[1] & kLinfoLenMask) == (infov[1] >> 28);
      const u64 pm0 = __builtin_amdgcn_ballot_w64(inl && e1);              // (+) half changes (or neither)
      const u64 pm1 = __builtin_amdgcn_ballot_w64(inl && e0 && !e1);       // (-) half changes
      const u64 pm2 = __builtin_amdgcn_ballot_w64(inl && !e0 && !e1);      // both change
      const u32x8 GCRE_CONSTANT* slots = (const u32x8 GCRE_CONSTANT*)(a.dlist + (u64)first * 16u);

      // ---- base counters of the two halves: the planes of paths0[row0] -- stored, or (REC) rebuilt from the recipe of the
      // join that produced the row: planes of ITS paths0 row, per half +/- what it added ----
      u32 B[2][LP];
      if constexpr (!REC) {
#pragma unroll
        for (int h = 0; h < 2; h++)
          load_groups(B[h], a.planes0, ((u64)kt * (u64)a.rows0 + (u64)row0 * 2 + h) * (u64)a.g0, a.g0);
      } else {
        const u32 GCRE_CONSTANT* r_row0 = (const u32 GCRE_CONSTANT*)a.rec_row0;
        const u32 GCRE_CONSTANT* r_rowz = (const u32 GCRE_CONSTANT*)a.rec_rowz;
        const u32 GCRE_CONSTANT* r_linfo = (const u32 GCRE_CONSTANT*)a.rec_linfo;
        const u32 GCRE_CONSTANT* r_lover = (const u32 GCRE_CONSTANT*)a.rec_lover;
        const u32 ra = r_row0[row0], rzr = r_rowz[row0];
        const u32 ri0 = r_linfo[(u64)row0 * 2], ri1 = r_linfo[(u64)row0 * 2 + 1];
#pragma unroll
        for (int h = 0; h < 2; h++)
          load_groups(B[h], a.rec_planes_a, ((u64)kt * (u64)a.rec_rows_a + (u64)ra * 2 + h) * (u64)a.rec_ga, a.rec_ga);
#pragma unroll
        for (int h = 0; h < 2; h++) {
          const u32 rinfo = h ? ri1 : ri0;
          const u32 rlen = rinfo & kLinfoLenMask;
          if (!(rinfo & 1u) && rlen == (rinfo >> 28)) continue;   // the producing join left this half as it was
          const u32x8 ro = *(const u32x8 GCRE_CONSTANT*)(a.rec_slot + ((u64)row0 * 2 + h) * 8u);
          u32 yr[8];
#pragma unroll
          for (int j = 0; j < 8; j++) yr[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, ro[j], 0);
          u32 ZR[LP];
          if (rinfo & 1u) {
            const u32 hz = (rzr >> 31) ? (u32)(1 - h) : (u32)h;
            load_groups(ZR, a.rec_planes_z, ((u64)kt * (u64)a.rec_rows_z + (u64)(rzr & 0x7fffffffu) * 2 + hz) * (u64)a.rec_gz, a.rec_gz);
          }
          u32 S[L];
          list_sum(S, yr, rlen, (const u32 GCRE_CONSTANT*)(a.rec_over + r_lover[(u64)row0 * 2 + h]));
          if (rinfo & 1u) {   // B = A + Z - S
            u32 cy = 0u, bw = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 s1_ = xor3(B[h][l], ZR[l], cy);
              cy = majority(B[h][l], ZR[l], cy);
              B[h][l] = xor3(s1_, S[l], bw);
              bw = borrow3(s1_, S[l], bw);
            }
          } else {            // B = A + S
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 bl = B[h][l];
              B[h][l] = xor3(bl, S[l], cy);
              cy = majority(bl, S[l], cy);
            }
          }
        }
      }
      auto write_out = [&](u32 h, u32 t, const auto& C) {   // planes of half h of joined path first + t
        if constexpr (OUT) {
          const u64 rh = ((u64)a.out_first + first + t) * 2 + h;
          u32x4* dst = (u32x4*)(a.planes_out + (((u64)kt * (u64)a.rows_out + rh) * (u64)a.go) * 256u) + lane;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (j < a.go) {
              u32x4 v = {0u, 0u, 0u, 0u};
              if (4 * j < L) v = u32x4{C[(4 * j) % L], (4 * j + 1 < L) ? C[(4 * j + 1) % L] : 0u,
                                       (4 * j + 2 < L) ? C[(4 * j + 2) % L] : 0u, (4 * j + 3 < L) ? C[(4 * j + 3) % L] : 0u};
              dst[j * 64] = v;
            }
          }
        }
      };
      u32 yA[8], yB[8], ZA[LZ], ZB[LZ];
#pragma unroll
      for (int l = 0; l < LZ; l++) ZA[l] = ZB[l] = 0u;

      // ---- both halves change (level 5, joins without a reduced operand): half by half, one half ahead -- the rows and
      // planes of the next half are in flight while this one is added up; (+) halves use buffer A, (-) halves buffer B ----
      if (pm2 != 0ull) {
        u64 rem = pm2;
        u32 t = (u32)__builtin_ctzll(rem);
        rem &= rem - 1ull;
        u32x8 oA = slots[t * 2u], oB = slots[t * 2u + 1u];
        issue(rdlane(infov[0], t), rdlane(zunit[0], t), oA, yA, ZA);
        for (;;) {
          issue(rdlane(infov[1], t), rdlane(zunit[1], t), oB, yB, ZB);
          const bool more = rem != 0ull;
          const u32 tn = more ? (u32)__builtin_ctzll(rem) : t;
          rem &= rem - 1ull;
          oA = slots[tn * 2u];
          u32 C0[L], C1[L];
          add_half(C0, B[0], yA, ZA, rdlane(infov[0], t), (const u32 GCRE_CONSTANT*)(a.dover + rdlane(lovv[0], t)));
          write_out(0u, t, C0);
          issue(rdlane(infov[0], tn), rdlane(zunit[0], tn), oA, yA, ZA);   // (the last path is simply requested twice)
          oB = slots[tn * 2u + 1u];
          add_half(C1, B[1], yB, ZB, rdlane(infov[1], t), (const u32 GCRE_CONSTANT*)(a.dover + rdlane(lovv[1], t)));
          write_out(1u, t, C1);
          // safe when F <= r_s and G <= r_{NS-1-s} for some step s
          u32 m = valid;
#pragma unroll
          for (int s = 0; s < NS; s++) m &= outside(C0, rdlane(ladv[s][0], t)) | outside(C1, rdlane(ladv[NS - 1 - s][1], t));
          if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
            n_slow++;
            enqueue(m, C0, C1, sp_diag_offset(rdlane(totv[0], t)), sp_diag_offset(rdlane(totv[1], t)));
          }
          if (!more) break;
          t = tn;
        }
      }

      // ---- one half changes: index 0 = the half that changes, index 1 = the half that is paths0's ----
#pragma nounroll
      for (u32 hc = 0u; hc < 2u; hc++) {
        const u64 pm = hc ? pm1 : pm0;
        if (pm == 0ull) continue;
        if (hc) {   // the (-) half's turn: the two halves trade places
#pragma unroll
          for (int l = 0; l < LP; l++) { const u32 x = B[0][l]; B[0][l] = B[1][l]; B[1][l] = x; }
          { const u32 x = infov[0]; infov[0] = infov[1]; infov[1] = x; }
          { const u32 x = lovv[0]; lovv[0] = lovv[1]; lovv[1] = x; }
          { const u32 x = zunit[0]; zunit[0] = zunit[1]; zunit[1] = x; }
          { const u32 x = totv[0]; totv[0] = totv[1]; totv[1] = x; }
#pragma unroll
          for (int s = 0; s < NS; s++) { const u32 x = ladv[s][0]; ladv[s][0] = ladv[s][1]; ladv[s][1] = x; }
        }
        // the unchanged half: the same counts, diagonal and intervals for every path of the class
        const u32 t0 = (u32)__builtin_ctzll(pm);
        u32 U[NS];
#pragma unroll
        for (int s = 0; s < NS; s++) U[s] = outside(B[1], rdlane(ladv[NS - 1 - s][1], t0));
        const u32 dU = sp_diag_offset(rdlane(totv[1], t0));
        auto compute = [&](u32 t, const u32 (&y)[8], const u32 (&Z)[LZ]) {
          u32 C[L];
          add_half(C, B[0], y, Z, rdlane(infov[0], t), (const u32 GCRE_CONSTANT*)(a.dover + rdlane(lovv[0], t)));
          write_out(hc, t, C);
          write_out(1u - hc, t, B[1]);
          u32 m = valid;
#pragma unroll
          for (int s = 0; s < NS; s++) m &= outside(C, rdlane(ladv[s][0], t)) | U[s];
          if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) return;
          n_slow++;
          enqueue(m, C, B[1], sp_diag_offset(rdlane(totv[0], t)), dU);
        };
        // one path ahead: the rows and planes of the next path are in flight while this one is computed
        u64 rem = pm;
        u32 tA = t0;
        rem &= rem - 1ull;
        bool hasB = rem != 0ull;
        u32 tB = hasB ? (u32)__builtin_ctzll(rem) : tA;
        rem &= rem - 1ull;
        u32x8 oA = slots[tA * 2u + hc], oB = slots[tB * 2u + hc];
        issue(rdlane(infov[0], tA), rdlane(zunit[0], tA), oA, yA, ZA);
        for (;;) {
          issue(rdlane(infov[0], tB), rdlane(zunit[0], tB), oB, yB, ZB);
          const bool hasC = rem != 0ull;
          const u32 tC = hasC ? (u32)__builtin_ctzll(rem) : tB;
          rem &= rem - 1ull;
          oA = slots[tC * 2u + hc];
          compute(tA, yA, ZA);
          if (!hasB) break;
          issue(rdlane(infov[0], tC), rdlane(zunit[0], tC), oA, yA, ZA);
          const bool hasD = rem != 0ull;
          const u32 tD = hasD ? (u32)__builtin_ctzll(rem) : tC;
          rem &= rem - 1ull;
          oB = slots[tD * 2u + hc];
          compute(tB, yB, ZB);
          if (!hasC) break;
          tA = tC;
          tB = tD;
          hasB = hasD;
        }
      }
    }
  }
  flush_tile();
  if (a.stats && lane == 0 && n_slow) atomicAdd(a.stats, n_slow);
}

// counter planes of the joined paths x plane groups of the added rows
#define GCRE_IE_M2P(EXPR)                                                                  \
  if (planes <= 8) { EXPR(8, 2); }                                                         \
  else if (planes <= 10) { if (a.gz <= 2) { EXPR(10, 2); } else { EXPR(10, 3); } }         \
  else if (planes <= 12) { EXPR(12, 3); }                                                  \
  else { EXPR(16, 4); }

#define GCRE_IE_M2_OR(EXPR, LL, GG)                                                            \
  if (out) { if (rec) { EXPR(LL, GG, true, true); } else { EXPR(LL, GG, true, false); } }       \
  else { if (rec) { EXPR(LL, GG, false, true); } else { EXPR(LL, GG, false, false); } }

hipError_t launch_null_ie_m2(const IeArgs& a, int planes, hipStream_t stream) {
  const dim3 grid((unsigned)(8 * a.waves_per_xcd / kIeWaves));
  const dim3 block(64 * kIeWaves);
  const bool out = a.planes_out != nullptr;
  const bool rec = a.rec_slot != nullptr;
#define GCRE_LAUNCH2X(LL, GG, OO, RR) hipLaunchKernelGGL((k_null_ie_m2<LL, GG, OO, RR>), grid, block, 0, stream, a)
#define GCRE_LAUNCH2(LL, GG) GCRE_IE_M2_OR(GCRE_LAUNCH2X, LL, GG)
  GCRE_IE_M2P(GCRE_LAUNCH2)
#undef GCRE_LAUNCH2
#undef GCRE_LAUNCH2X
  return hipGetLastError();
}

int ie2_max_waves_per_cu(int planes, int gz, bool out, bool rec) {
  int blocks = 0;
  hipError_t e = hipSuccess;
  struct { int gz; } a{gz};
#define GCRE_OCC2X(LL, GG, OO, RR) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_null_ie_m2<LL, GG, OO, RR>, 64 * kIeWaves, 0)
#define GCRE_OCC2(LL, GG) GCRE_IE_M2_OR(GCRE_OCC2X, LL, GG)
  GCRE_IE_M2P(GCRE_OCC2)
#undef GCRE_OCC2
#undef GCRE_OCC2X
  if (e != hipSuccess || blocks < 1) blocks = 1;
  return blocks * kIeWaves;
}

int ie2_steps() { return GCRE_M2_STEPS; }

}  // namespace gcre
