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
// * ONE interval test where the unchanged half cannot matter.  The steps' intervals are nested, so when no permutation of
//   the tile has the unchanged half's G above the lowest step, the widest interval of the changed half decides alone; and
//   when the unchanged half is EMPTY (about a third of the level-4 paths: nothing negative on the path), G is the constant
//   vtmax[0][0] and the changed half is tested against the interval of theta - G itself -- the method-1 test.
// * Kept rows leave a RECIPE here too (REC): row r = row rec_row0[r] of set A | row rec_rowz[r] of set Z, per half the
//   producing join's list.  The segment's base counters are rebuilt from the planes of A and Z in the prologue; the kept
//   set's own planes (68 GB per pass at configs[2] geometry) are neither written nor read.
#include "gcre_ie_common.h"

#ifndef GCRE_M2_STEPS
#define GCRE_M2_STEPS 2
#endif
#ifndef GCRE_M2_YW_WIDE
#define GCRE_M2_YW_WIDE 4     // mask rows fetched a path ahead in the variants with more than 10 planes (8: spills, configs[4] 324 instead of 291 ms)
#endif
#ifndef GCRE_M2_WAVES16
#define GCRE_M2_WAVES16 3     // waves per SIMD of the 16-plane variants (tuning)
#endif
#ifndef GCRE_M2_MODES
#define GCRE_M2_MODES 3   // bit 0: one test when the unchanged half is empty, bit 1: one test when its G stays under the lowest step
#endif

namespace gcre {

template <int L, int GZ, bool OUT, bool REC>
__global__ __launch_bounds__(64 * kIeWaves) __attribute__((amdgpu_waves_per_eu(L <= 10 ? 4 : (L <= 12 ? 3 : GCRE_M2_WAVES16)))) void k_null_ie_m2(const IeArgs a) {
  constexpr int NS = GCRE_M2_STEPS;
  static_assert(NS == 2 || NS == 3, "two or three steps");
  constexpr int LP = (L + 3) / 4 * 4;
  // mask rows of a list fetched a path ahead: four in the 4-wave variants (rare variants: most overlaps are shorter, and
  // four more registers in both buffers cost spills inside the path loop), the whole slot where there are registers
  constexpr int YW = L <= 10 ? 4 : GCRE_M2_YW_WIDE;
  constexpr int LZ = 4 * GZ;   // planes of an added row (GZ groups; the launch picks GZ >= a.gz)
  static_assert(LZ <= LP, "added rows have no more planes than joined paths");
  static_assert(L >= 8 && L <= 16, "8 to 16 counter planes");
  typedef u32 __attribute__((ext_vector_type(8))) u32x8;
  typedef u32 __attribute__((ext_vector_type(YW))) SlotV;   // the slot entries fetched a path ahead (scalar registers)
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
  // the row for a changed half whose other half is empty: G = vtmax[0][0] <= g00_rows / (2 kLadderPerUnit), so F may use
  // all of 2 j - g00_rows (top_ok: that is not negative)
  u32 lad_top = lad[0];
  bool top_ok = a.lad_mode != 0 || a.g00_rows == 0u;
  u32 n_slow = 0u;
#ifdef GCRE_IE_TIMING
  // diagnostics build: path-tiles by class -- [0] other half empty, [1] one test, [2] all steps, [3] both halves change,
  // [4] lists of 5..8 rows (fetched late), [5] permutations queued for a look-up, [6] long lists
  u64 tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define GCRE_TM_COUNT(i, n) tm[i] += (u64)(n)
#else
#define GCRE_TM_COUNT(i, n)
#endif
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
    const int rt = 2 * j - (int)a.g00_rows;
    top_ok = a.g00_rows != 0xffffffffu && rt >= 0;
    lad_top = (u32)(rt < 0 ? 0 : (rt > kLadder2Levels - 1 ? kLadder2Levels - 1 : rt)) * (u32)a.ladder_stride;
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
  // y[0] + y[1] + y[2] + y[3], bit-sliced: three planes
  auto sum4 = [&](const u32 (&y)[4], u32 (&S)[4]) {
    const u32 a0 = xor3(y[0], y[1], y[2]), c0 = majority(y[0], y[1], y[2]);
    S[0] = a0 ^ y[3];
    const u32 c1 = a0 & y[3];
    S[1] = c0 ^ c1;
    S[2] = c0 & c1;
    S[3] = 0u;
  };
  // S += the mask rows of a long list beyond its first 8 entries (blocks of 8; entries 8.. live at over + *lover_at)
  auto sum_more = [&](u32 (&S)[L], u32 len, const u32 GCRE_CONSTANT* over, const u32 GCRE_CONSTANT* lover_at) {
    const u32 GCRE_CONSTANT* more = over + *lover_at;
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
  // C = Bx + Nz - S (overlap list) or Bx + S (delta list).  info = the list's linfo word, y = its first four mask rows
  // (issue), slot_at = its 8-entry slot, lover_at = where its lover word sits (long lists)
  auto add_half = [&](u32 (&C)[L], const auto& Bx, const u32 (&y)[YW], const u32 (&Z)[LZ], u32 info, const u32 GCRE_CONSTANT* slot_at,
                      const u32 GCRE_CONSTANT* lover_at) {
    const u32 len = info & kLinfoLenMask;
    const bool overlap = (info & 1u) != 0u;
    const u32 real = len - (info >> 28);
    u32 S4[4];
    {
      const u32 y03[4] = {y[0], y[1], y[2], y[3]};
      sum4(y03, S4);
    }
    if (real > 4u) {
      GCRE_TM_COUNT(4, 1);
      u32 y2[4], s2[4];
      if constexpr (YW == 8) {
#pragma unroll
        for (int j = 0; j < 4; j++) y2[j] = y[(4 + j) % YW];
      } else {
        // entries 5..8 of the slot: fetched here, not a path ahead
        const u32x4 o = *(const u32x4 GCRE_CONSTANT*)(slot_at + 4);
#pragma unroll
        for (int j = 0; j < 4; j++) y2[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, o[j], 0);
      }
      sum4(y2, s2);
      const u32 t0 = S4[0] ^ s2[0], c0 = S4[0] & s2[0];
      const u32 t1 = xor3(S4[1], s2[1], c0), c1 = majority(S4[1], s2[1], c0);
      const u32 t2 = xor3(S4[2], s2[2], c1), c2 = majority(S4[2], s2[2], c1);
      S4[0] = t0; S4[1] = t1; S4[2] = t2; S4[3] = c2;
    }
    if (len <= 8u) {
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
    GCRE_TM_COUNT(6, 1);
    u32 S[L];
#pragma unroll
    for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
    sum_more(S, len, (const u32 GCRE_CONSTANT*)a.dover, lover_at);
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
  // the loads of one list a path ahead: its first four mask rows (none when the list is empty) and, for an overlap list,
  // the planes of the added row-half
  auto issue = [&](u32 info, u32 zu, const SlotV offs, u32 (&yy)[YW], u32 (&ZZ)[LZ]) {
    const u32 real = (info & kLinfoLenMask) - (info >> 28);
    if (YW == 8 && real > 4u) {
#pragma unroll
      for (int j = 0; j < 8; j++) yy[j % YW] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j % YW], 0);
    } else if (real > 0u) {
#ifdef GCRE_M2_NOROWS   // timing experiment (wrong results): no mask-row loads
#pragma unroll
      for (int j = 0; j < 4; j++) yy[j] = offs[j];
#else
#pragma unroll
      for (int j = 0; j < 4; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, offs[j], 0);
#endif
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) yy[j] = 0u;
    }
    if (info & 1u) {
      __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(k_planesz + (u64)zu * 1024u), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < GZ; j++) {
        u32x4 v = {0u, 0u, 0u, 0u};
#ifdef GCRE_M2_NOZLOAD   // timing experiment (wrong results): no plane loads of the added rows
        v = u32x4{zu, zu + (u32)j, lane4, 0u};
#else
        if (j < 2 || j < a.gz) v = __builtin_amdgcn_raw_buffer_load_b128(rz, lane4 * 4u + (u32)j * 1024u, 0, 0);   // (a.gz >= 2: plane_groups_for)
#endif
        ZZ[4 * j + 0] = v.x; ZZ[4 * j + 1] = v.y; ZZ[4 * j + 2] = v.z; ZZ[4 * j + 3] = v.w;
      }
    }
  };
  // the permutations in m leave their two cells and their slot in the look-up queue, one per lane and round
  auto enqueue = [&](u32 m, const auto& Ca, const auto& Cb, u32 da, u32 db, bool b_zero) {
    while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
      if (lq_n + 64u > kLqCap) lq_drain();
      const bool has = m != 0u;
      const u32 bb = has ? (u32)__builtin_ctz(m) : 0u;
      m &= m - 1u;
      u32 ca = 0u, cb = 0u;   // the permutation's two counts, top plane first: v_bfe_u32 + v_lshl_or_b32 per plane
#pragma unroll
      for (int l = L - 1; l >= 0; l--) ca = (ca << 1) | __builtin_amdgcn_ubfe(Ca[l], bb, 1u);
      if (!b_zero) {
#pragma unroll
        for (int l = L - 1; l >= 0; l--) cb = (cb << 1) | __builtin_amdgcn_ubfe(Cb[l], bb, 1u);
      }
      const u64 hm = __builtin_amdgcn_ballot_w64(has);
      GCRE_TM_COUNT(5, __builtin_popcountll(hm));
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
        lad_top = 0u;
        top_ok = a.g00_rows == 0u;
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
      // ---- per-path metadata of the segment: lane t <-> joined path first + t.  A half with an empty delta list is the
      // paths0 half (the added row has nothing there, or nothing new): for the paths with one such half only the OTHER
      // half's list info, planes and ladder rows stay in vector registers ----
      const bool inl = (u32)lane < npaths;
      const u32 qv = first + (inl ? (u32)lane : 0u);
      const u32 rzv = a.rowz[qv];
      const bool mine = (u64)sidx < (u64)a.score_segs;   // else: rows of another shard, the ladder's all-inside row
      const bool seg_top_ok = top_ok;   // the rows read below and this flag belong together (an exchange moves both)
      // (where a long list continues, the slot entries and the carrier totals are read where they are used, through the
      // scalar cache)
      const u32 GCRE_CONSTANT* k_lover = (const u32 GCRE_CONSTANT*)a.lover + (u64)first * 2u;
      const u32 GCRE_CONSTANT* k_tot = (const u32 GCRE_CONSTANT*)a.tot + (u64)first * 2u;
      const u32 GCRE_CONSTANT* k_slot = (const u32 GCRE_CONSTANT*)a.dlist + (u64)first * 16u;
      u32 infoc, zunitc, ladc[NS], ladt, ladu[NS], totu;
      u64 pm0, pm1, pm2;
      {
        const u32 i0 = a.linfo[(u64)qv * 2], i1 = a.linfo[(u64)qv * 2 + 1];
        const u32 t0_ = a.tot[(u64)qv * 2], t1_ = a.tot[(u64)qv * 2 + 1];
        const bool e0 = !(i0 & 1u) && (i0 & kLinfoLenMask) == (i0 >> 28);
        const bool e1 = !(i1 & 1u) && (i1 & kLinfoLenMask) == (i1 >> 28);
        pm0 = __builtin_amdgcn_ballot_w64(inl && e1);              // the (+) half changes (or neither)
        pm1 = __builtin_amdgcn_ballot_w64(inl && e0 && !e1);       // the (-) half changes
        pm2 = __builtin_amdgcn_ballot_w64(inl && !e0 && !e1);      // both change
        const bool c1 = e0 && !e1;
        infoc = c1 ? i1 : i0;
        const u32 hz = ((rzv >> 31) != 0u) != c1 ? 1u : 0u;        // the added row's half that lands in the changed half
        zunitc = ((u32)kt * (u32)a.rowsz + (rzv & 0x7fffffffu) * 2u + hz) * (u32)a.gz;
        const u32 tc = c1 ? t1_ : t0_;
        totu = c1 ? t0_ : t1_;
#pragma unroll
        for (int s = 0; s < NS; s++) {
          ladc[s] = a.ladder[(mine ? lad[s] : lad_keep) + tc];
          ladu[s] = a.ladder[(mine ? lad[NS - 1 - s] : lad_keep) + totu];   // step s pairs F <= r_s with G <= r_{NS-1-s}
        }
        ladt = a.ladder[(mine ? lad_top : lad_keep) + tc];   // the changed half's interval when the other half is empty
      }

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
          {
            u32 S4[4];
            sum8(yr, S4);
#pragma unroll
            for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
          }
          if (rlen > 8u) sum_more(S, rlen, (const u32 GCRE_CONSTANT*)a.rec_over, r_lover + (u64)row0 * 2 + h);
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
      u32 yA[YW], yB[YW], ZA[LZ], ZB[LZ];
#pragma unroll
      for (int l = 0; l < LZ; l++) ZA[l] = ZB[l] = 0u;

      // ---- both halves change (level 5, joins without a reduced operand): half by half, one half ahead -- the rows and
      // planes of the next half are in flight while this one is added up; (+) halves use buffer A, (-) halves buffer B.
      // These paths keep both halves' words per lane (loaded here: segments without such paths never pay for them) ----
      if (pm2 != 0ull) {
        u32 info2[2], zunit2[2], lad2[NS][2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
          info2[h] = a.linfo[(u64)qv * 2 + h];
          const u32 hz = (rzv >> 31) ? (u32)(1 - h) : (u32)h;
          zunit2[h] = ((u32)kt * (u32)a.rowsz + (rzv & 0x7fffffffu) * 2u + hz) * (u32)a.gz;
          const u32 th = a.tot[(u64)qv * 2 + h];
#pragma unroll
          for (int s = 0; s < NS; s++) lad2[s][h] = a.ladder[(mine ? lad[s] : lad_keep) + th];
        }
        u64 rem = pm2;
        u32 t = (u32)__builtin_ctzll(rem);
        rem &= rem - 1ull;
        SlotV oA = *(const SlotV GCRE_CONSTANT*)(k_slot + t * 16u), oB = *(const SlotV GCRE_CONSTANT*)(k_slot + t * 16u + 8u);
        issue(rdlane(info2[0], t), rdlane(zunit2[0], t), oA, yA, ZA);
        for (;;) {
          issue(rdlane(info2[1], t), rdlane(zunit2[1], t), oB, yB, ZB);
          const bool more = rem != 0ull;
          const u32 tn = more ? (u32)__builtin_ctzll(rem) : t;
          rem &= rem - 1ull;
          oA = *(const SlotV GCRE_CONSTANT*)(k_slot + tn * 16u);
          u32 C0[L], C1[L];
          add_half(C0, B[0], yA, ZA, rdlane(info2[0], t), k_slot + t * 16u, k_lover + t * 2u);
          write_out(0u, t, C0);
          issue(rdlane(info2[0], tn), rdlane(zunit2[0], tn), oA, yA, ZA);   // (the last path is simply requested twice)
          oB = *(const SlotV GCRE_CONSTANT*)(k_slot + tn * 16u + 8u);
          add_half(C1, B[1], yB, ZB, rdlane(info2[1], t), k_slot + t * 16u + 8u, k_lover + t * 2u + 1u);
          write_out(1u, t, C1);
          // safe when F <= r_s and G <= r_{NS-1-s} for some step s
          u32 m = valid;
#pragma unroll
          for (int s = 0; s < NS; s++) m &= outside(C0, rdlane(lad2[s][0], t)) | outside(C1, rdlane(lad2[NS - 1 - s][1], t));
          GCRE_TM_COUNT(3, 1);
          if (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {
            n_slow++;
            enqueue(m, C0, C1, sp_diag_offset(k_tot[t * 2u]), sp_diag_offset(k_tot[t * 2u + 1u]), false);
          }
          if (!more) break;
          t = tn;
        }
      }

      // ---- one half changes.  What concerns the unchanged half is the same for every path of a class -- its counts (the
      // base counters), its diagonal, its interval tests U[s] -- and is settled here for both classes ----
      const u32 t00 = pm0 ? (u32)__builtin_ctzll(pm0) : 0u, t01 = pm1 ? (u32)__builtin_ctzll(pm1) : 0u;
      const u32 totU0 = rdlane(totu, t00), totU1 = rdlane(totu, t01);
      // empty: its G is the constant vtmax[0][0], the changed half has the rest of theta to itself -- one test
      const bool empty0 = (GCRE_M2_MODES & 1) && pm0 != 0ull && totU0 == 0u && seg_top_ok;
      const bool empty1 = (GCRE_M2_MODES & 1) && pm1 != 0ull && totU1 == 0u && seg_top_ok;
      u32 U[NS], U1[NS];
#pragma unroll
      for (int s = 0; s < NS; s++) U[s] = U1[s] = 0u;
      bool single0 = true, single1 = true;
      if (pm0 != 0ull && !empty0) {
#pragma unroll
        for (int s = 0; s < NS; s++) U[s] = outside(B[1], rdlane(ladu[s], t00));
        // no live permutation has G above the lowest step: U[s] lies inside U[NS - 1] for every s and the changed half's
        // intervals are nested, so the widest one decides alone
        single0 = (GCRE_M2_MODES & 2) && __builtin_amdgcn_ballot_w64((U[NS - 1] & valid) != 0u) == 0ull;
      }
      if (pm1 != 0ull && !empty1) {
#pragma unroll
        for (int s = 0; s < NS; s++) U1[s] = outside(B[0], rdlane(ladu[s], t01));
        single1 = (GCRE_M2_MODES & 2) && __builtin_amdgcn_ballot_w64((U1[NS - 1] & valid) != 0u) == 0ull;
      }
      // index 0 = the half that changes, index 1 = the half that is paths0's: for the (-) half's turn the base counters
      // trade places, so that one loop body serves both orientations
#pragma nounroll
      for (u32 hc = 0u; hc < 2u; hc++) {
        const u64 pm = hc ? pm1 : pm0;
        if (pm == 0ull) continue;
        if (hc) {
#pragma unroll
          for (int l = 0; l < L; l++) { const u32 x = B[0][l]; B[0][l] = B[1][l]; B[1][l] = x; }
#pragma unroll
          for (int s = 0; s < NS; s++) U[s] = U1[s];
        }
        const bool u_empty = hc ? empty1 : empty0;
        const bool single = hc ? single1 : single0;
        const u32 dU = sp_diag_offset(hc ? totU1 : totU0);
        const u32 ladw = u_empty ? ladt : ladc[NS - 1];   // the widest interval of the changed half
        auto compute = [&](u32 t, const u32 (&y)[YW], const u32 (&Z)[LZ]) {
          u32 C[L];
          add_half(C, B[0], y, Z, rdlane(infoc, t), k_slot + t * 16u + hc * 8u, k_lover + t * 2u + hc);
          write_out(hc, t, C);
          write_out(1u - hc, t, B[1]);
          GCRE_TM_COUNT(u_empty ? 0 : (single ? 1 : 2), 1);
          u32 m = outside(C, rdlane(ladw, t));
          if (!single) {
            // (the barrier keeps the compiler from running these tests for every path and selecting afterwards)
            asm volatile("" ::: "memory");
            m |= U[NS - 1];
#pragma unroll
            for (int s = 0; s < NS - 1; s++) m &= outside(C, rdlane(ladc[s], t)) | U[s];
          }
          m &= valid;
          if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) return;
          n_slow++;
#ifdef GCRE_M2_NOLOOKUP   // timing experiment (wrong results): the tests run, nothing is looked up
          return;
#endif
          enqueue(m, C, B[1], sp_diag_offset(k_tot[t * 2u + hc]), dU, u_empty);
        };
        // one path ahead: the rows and planes of the next path are in flight while this one is computed
        u64 rem = pm;
        u32 tA = (u32)__builtin_ctzll(rem);
        rem &= rem - 1ull;
        bool hasB = rem != 0ull;
        u32 tB = hasB ? (u32)__builtin_ctzll(rem) : tA;
        rem &= rem - 1ull;
        // (the slot's entries come through the scalar cache: requested one path before the loads that use them)
        SlotV oA = *(const SlotV GCRE_CONSTANT*)(k_slot + tA * 16u + hc * 8u), oB = *(const SlotV GCRE_CONSTANT*)(k_slot + tB * 16u + hc * 8u);
        issue(rdlane(infoc, tA), rdlane(zunitc, tA), oA, yA, ZA);
        for (;;) {
          issue(rdlane(infoc, tB), rdlane(zunitc, tB), oB, yB, ZB);
          const bool hasC = rem != 0ull;
          const u32 tC = hasC ? (u32)__builtin_ctzll(rem) : tB;
          rem &= rem - 1ull;
          oA = *(const SlotV GCRE_CONSTANT*)(k_slot + tC * 16u + hc * 8u);
          compute(tA, yA, ZA);
          if (!hasB) break;
          issue(rdlane(infoc, tC), rdlane(zunitc, tC), oA, yA, ZA);
          const bool hasD = rem != 0ull;
          const u32 tD = hasD ? (u32)__builtin_ctzll(rem) : tC;
          rem &= rem - 1ull;
          oB = *(const SlotV GCRE_CONSTANT*)(k_slot + tD * 16u + hc * 8u);
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
#ifdef GCRE_IE_TIMING
  if (a.timing && lane == 0)
    for (int i = 0; i < 7; i++) atomicAdd((unsigned long long*)a.timing + i, (unsigned long long)tm[i]);
#endif
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
