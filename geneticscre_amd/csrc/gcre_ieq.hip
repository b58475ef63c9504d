// gcre_ieq.hip -- the pruned method-1 null kernel, "quad" form: one wave scores FOUR joined paths at a time.
//
// Same arithmetic as k_null_ie_m1 (gcre_ie.hip): for every joined path and 2048-permutation tile
//
//     count = N0[idx] + Nz[z] - popc(p0[idx] & z & mask_r)        (bit-sliced, reference src/methods.h:73-88)
//
// followed by the exact interval test against the pruning ladder and table look-ups (methods.h:96-103) only for the
// permutations that fail it.  What changes is how the operands reach the lanes.  k_null_ie_m1 gives every lane 32
// permutations of ONE path, so each of the 8 mask rows of a path is a 256-byte wave load of 4 bytes per lane -- and a
// CU's vector-memory pipe takes ~15 clocks per wave-load whatever its width (tools/row_gather_rate.hip): 8 row loads +
// 2 plane loads per path-tile = 146 CU-clocks, the measured cost of that kernel.  It is bound by the number of
// vector-memory INSTRUCTIONS, not by bytes, VALU issue or latency.
//
// Here a wave is four groups of 16 lanes.  Group g works on its own uid (segment); lane (g, s) holds FOUR dwords --
// 128 permutations -- of every row and plane of its group's path: dwords w = s + 16 d, d = 0..3, of the 64-dword tile.
//   * mask rows: one buffer_load_dwordx4 fetches one row for each of the four paths (16 lanes x 16 B = 256 B per group,
//     from a copy of the transposed masks whose dwords are stored in lane order, `mtq`): 8 loads per FOUR path-tiles,
//     ~19 clocks each instead of 4 x 15;
//   * added rows: the four segments of a quad join the SAME paths1 rows (all uids with one pivot gene do: the host
//     groups them, gcre_host.hip quad table), so the planes of the added row are loaded once per wave (GZ wide loads)
//     and handed to the four groups through LDS;
//   * per-path metadata (list slot, list info, carrier total, ladder entry, added row) is staged through LDS sixteen
//     paths at a time, read back with broadcast ds_read_b128 -- nothing is moved through scalar registers.
// VALU work per path-tile is unchanged (every instruction still processes 64 x 32 counters); vector-memory
// instructions drop from 13 to ~3 per path-tile.
#include "gcre_ie_common.h"

namespace gcre {

constexpr int kQChunk = 16;       // joined paths (per group) whose metadata is staged at a time
constexpr int kQMeta = 12;        // words per staged path: slot[8], linfo, tot, lover, added row
constexpr int kQS = 6;            // planes of a list's row sum: lists of up to 56 entries (the host sends longer ones to k_null_ie_m1)

typedef u32 __attribute__((ext_vector_type(3))) u32x3;

// (a & m) | (b & ~m)
__device__ __forceinline__ u32 mux3(u32 a, u32 b, u32 m) { return __builtin_amdgcn_bitop3_b32(a, b, m, 0xE4); }

// -DGCRE_IE_TIMING: per-section s_memtime sums (no extra waits: a mark only reads the clock)
#ifdef GCRE_IE_TIMING
#define GCRE_QT(var) const u64 var = __builtin_amdgcn_s_memtime()
#define GCRE_QT_ADD(i, t1, t0) tm[i] += (t1) - (t0)
#else
#define GCRE_QT(var)
#define GCRE_QT_ADD(i, t1, t0)
#endif

template <int L, int GZ, bool OUT, bool REC>
__global__ __launch_bounds__(64 * kIeWaves) __attribute__((amdgpu_waves_per_eu(2))) void k_null_ie_q(const IeArgs a) {
  constexpr int LP = (L + 3) / 4 * 4;
  constexpr int LZ = 4 * GZ;
  static_assert(L >= 8 && L <= 16 && GZ >= 2 && LZ <= LP, "planes come in groups of 4");
  __shared__ u32 nmax_lds[kIeWaves][32 * 64];                       // running maxima [bit][dword] per wave
  __shared__ u32x4 z_lds[kIeWaves][GZ * 64];                        // planes of the added row of the current iteration
  __shared__ u32x4 meta_lds[kIeWaves][64 * kQMeta / 4];             // [group][path of the chunk][kQMeta words]
  __shared__ u32 wq_state[kIeWaves][8];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = lane >> 4, sub = lane & 15;
  // Launch constants that steer control flow or end up in buffer descriptors, pinned to scalar registers here: the
  // compiler re-loads kernel arguments on both sides of the `lane == 0` branch around the ticket atomic, and whatever is
  // derived from such a pair counts as divergent (waterfall loops around every buffer load, vector-register loop counters).
  auto uni = [](u32 v) -> u32 { return (u32)__builtin_amdgcn_readfirstlane((int)v); };
  const u32 k_quad_begin = uni((u32)a.quad_begin), k_quad_end = uni((u32)a.quad_end), k_batch = uni((u32)a.batch);
  const u32 k_nkt = uni((u32)a.nkt), k_K = uni((u32)a.K), k_mt_rows = uni(a.mt_rows), k_lstride = uni((u32)a.ladder_stride);
  const u32 k_lad_mode = uni((u32)a.lad_mode), k_score_segs = uni(a.score_segs);
  const u32* k_mtq = (const u32*)(((u64)uni((u32)((u64)a.mtq >> 32)) << 32) | (u64)uni((u32)(u64)a.mtq));
  u32* nm = nmax_lds[wave] + lane;
#pragma unroll
  for (int q = 0; q < 32; q++) nm[q * 64] = 0u;
  u32x4* zst = z_lds[wave];
  u32x4* mst = meta_lds[wave];
  const SparseSeg* segs = a.segs;

  int cur_kt = -1;
  u32 valid = 0u;          // exchange(): lane = dword `lane` of the tile
  bool tail_tile = false;  // the tile holds fewer than 2048 live permutations
  u32 lad_base = (k_lad_mode == 0u) ? 0u : ((u32)kLadderLevels - 1u + k_lad_mode) * k_lstride;
  const u32 lad_keep = (u32)kLadderLevels * k_lstride;
  bool dirty = false;
  u32 n_slow = 0u;
#ifdef GCRE_IE_TIMING
  u64 tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // quad header + base counters, row sums (incl. the wait for the rows), long lists, fetch, counts + test, look-ups, total, iterations
  const u64 tm_begin = __builtin_amdgcn_s_memtime();
#endif
  __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)k_mtq, 0, 0x7fffffff, 0x00020000);
  const u32 sub16 = (u32)sub * 16u;

  auto exchange = [&]() {
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
    int j = (int)(__uint_as_float(theta) * (float)kLadderPerUnit);
    j = j < 0 ? 0 : (j > kLadderLevels - 1 ? kLadderLevels - 1 : j);
    lad_base = (u32)j * k_lstride;
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

  WorkQueue wq;
  wq.st = wq_state[wave];
  wq.init(a.queue, (k_quad_end - k_quad_begin + k_batch - 1u) / k_batch, k_nkt);
  wq.select(blockIdx.x & 7u);
  u32 ticket = wq.take(lane);
  int since = 0, period = 1;
  for (;;) {
    const u32 work = __builtin_amdgcn_readfirstlane(ticket);
    const u32 q_n = wq.get(2);
    if (work >= q_n) {
      if (!wq.steal(lane)) break;
      ticket = wq.take(lane);
      continue;
    }
    ticket = wq.take(lane);
    const u32 item = wq.get(1) + work, nbt = wq.get(3);
    const int kt = __builtin_amdgcn_readfirstlane((int)(item / nbt));
    const u32 q_lo = uni(k_quad_begin + (item - (u32)kt * nbt) * k_batch);
    const u32 q_hi = uni(q_lo + k_batch < k_quad_end ? q_lo + k_batch : k_quad_end);
    if (kt != cur_kt) {
      flush_tile();
      cur_kt = kt;
      mt = __builtin_amdgcn_make_buffer_rsrc((void*)(k_mtq + (size_t)kt * k_mt_rows * 64), 0, 0x7fffffff, 0x00020000);
      const int live = (int)k_K - kt * 2048 - lane * 32;
      valid = live >= 32 ? 0xffffffffu : (live <= 0 ? 0u : ((1u << live) - 1u));
      tail_tile = (int)k_K - kt * 2048 < 2048;
      if (k_lad_mode == 0u) lad_base = 0u;
      since = 0;
      period = 1;
    }
    for (u32 qi = q_lo; qi < q_hi; qi++) {
      // ---- the quad: up to four consecutive segments that join the same paths1 rows; spare groups shadow the last one
      GCRE_QT(tq0);
      const u32 qe = a.quads[qi];
      const u32 qcnt = (qe >> 30) + 1u;
      const u32 sidx = (qe & 0x3fffffffu) + ((u32)grp < qcnt ? (u32)grp : qcnt - 1u);
      const u32* sgp = (const u32*)segs + (u64)sidx * 3u;
      const u32 row0 = sgp[0], first = sgp[1];
      const u32 npaths = __builtin_amdgcn_readfirstlane(sgp[2]);   // the segments of a quad have the same length
      u32x4 rc0 = {0u, 0u, 0u, 0u}, rc1 = rc0, rc2 = rc0;
      if constexpr (REC) {
        const u32x4* rs = (const u32x4*)(a.rec_segs + (u64)sidx * kRecSegWords);
        rc0 = rs[0];   // paths0 row of the producing join, row it added, list info, where a long list continues
        rc1 = rs[1];   // its slot
        rc2 = rs[2];
      }
      if (k_lad_mode == 0u && ++since >= period) {
        exchange();
        since = 0;
        period = period < kIeRefresh ? period * 2 : kIeRefresh;
      }
      const u32 lad_row = (sidx < k_score_segs) ? lad_base : lad_keep;
      const u32 last = npaths - 1u;

      // ---- metadata of 16 paths per group: lane (g, s) fetches path first_g + j0 + s ----
      auto stage_load = [&](u32 j0, u32 (&M)[kQMeta]) {
        const u32 jj = j0 + (u32)sub;
        const u64 p = (u64)first + (jj < last ? jj : last);
        const u32x4* sl = (const u32x4*)(a.dlist + p * 8u);
        const u32x4 e0 = sl[0], e1 = sl[1];
        M[0] = e0.x; M[1] = e0.y; M[2] = e0.z; M[3] = e0.w;
        M[4] = e1.x; M[5] = e1.y; M[6] = e1.z; M[7] = e1.w;
        M[8] = a.linfo[p];
        M[9] = a.tot[p];
        M[10] = a.lover[p];
        M[11] = a.rowz[p] & 0x7fffffffu;
      };
      auto stage_store = [&](u32 (&M)[kQMeta]) {
        u32x4* dst = mst + lane * (kQMeta / 4);
        dst[0] = u32x4{M[0], M[1], M[2], M[3]};
        dst[1] = u32x4{M[4], M[5], M[6], M[7]};
        dst[2] = u32x4{M[8], M[9], M[10], M[11]};
      };
      // What the lanes of a group need of their path j.  Everything that has to come from memory for it goes out one path
      // ahead: the 8 rows of its slot, the added row's planes, its ladder entry and -- when its list is longer than the
      // slot -- the next 8 entries of the list.
      struct PathMeta { u32 info, tot, lh, rowz, lov; u32x4 e2a, e2b; };
      auto fetch = [&](u32 jn, PathMeta& pm, u32x4 (&yy)[8], u32x4 (&zz)[GZ]) {
        const u32 jc = jn < last ? jn : last;   // past the end: the last path is simply requested again
        const u32x4* src = mst + (grp * kQChunk + (int)(jc & (kQChunk - 1))) * (kQMeta / 4);
        const u32x4 e0 = src[0], e1 = src[1], e2 = src[2];
        pm.info = e2.x; pm.tot = e2.y; pm.lov = e2.z; pm.rowz = e2.w;
        const u32 offs[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
#pragma unroll
        for (int k = 0; k < 8; k++) yy[k] = __builtin_amdgcn_raw_buffer_load_b128(mt, offs[k] + sub16, 0, 0);
        // the added row is the same for the four groups: group 0's copy of its number addresses the (wave-wide) loads
        const u32 zunit = ((u32)kt * (u32)a.rowsz + __builtin_amdgcn_readfirstlane(pm.rowz)) * (u32)a.gz;
        const u32x4* zsrc = (const u32x4*)(a.planesz + (u64)zunit * 256u) + lane;
#pragma unroll
        for (int jz = 0; jz < GZ; jz++) zz[jz] = zsrc[jz * 64];
        pm.lh = a.ladder[lad_row + pm.tot];
        pm.e2a = pm.e2b = u32x4{a.zoff, a.zoff, a.zoff, a.zoff};
        if (__builtin_amdgcn_ballot_w64((pm.info & kLinfoLenMask) > 8u) != 0ull) {   // somebody's list goes on (lists start 32-byte aligned)
          const u32x4* more = (const u32x4*)(a.dover + ((pm.info & kLinfoLenMask) > 8u ? pm.lov : 0u));
          const u32x4 ma = more[0], mb = more[1];
          if ((pm.info & kLinfoLenMask) > 8u) { pm.e2a = ma; pm.e2b = mb; }
        }
      };
      // the rest of a list: 8 more rows per round, every round summed into the planes S; groups that are through add the
      // all-zero row.  y2 = the rows of entries 8..15, already requested by the caller (or nullptr: fetch them here)
      auto add_blocks = [&](u32 (&S)[kQS][4], u32 len, u32 maxlen, const u32* over, u32 lov, const u32x4 (*y2p)[8]) {
        for (u32 q = 8u; q < maxlen; q += 8u) {
          u32x4 y2[8];
          if (q == 8u && y2p) {
#pragma unroll
            for (int k = 0; k < 8; k++) y2[k] = (*y2p)[k];
          } else {
            const u32x4* more = (const u32x4*)(over + (q < len ? lov + q - 8u : 0u));
            const u32x4 ma = more[0], mb = more[1];
            const bool in = q < len;
            const u32 o8[8] = {in ? ma.x : a.zoff, in ? ma.y : a.zoff, in ? ma.z : a.zoff, in ? ma.w : a.zoff,
                               in ? mb.x : a.zoff, in ? mb.y : a.zoff, in ? mb.z : a.zoff, in ? mb.w : a.zoff};
#pragma unroll
            for (int k = 0; k < 8; k++) y2[k] = __builtin_amdgcn_raw_buffer_load_b128(mt, o8[k] + sub16, 0, 0);
          }
#pragma unroll
          for (int d = 0; d < 4; d++) {
            u32 r8[8], s4[4];
#pragma unroll
            for (int k = 0; k < 8; k++) r8[k] = y2[k][d];
            sum8(r8, s4);
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < kQS; l++) {
              const u32 sv = S[l][d];
              if (l < 4) {
                S[l][d] = xor3(sv, s4[l < 4 ? l : 0], cy);
                cy = majority(sv, s4[l < 4 ? l : 0], cy);
              } else {
                S[l][d] = sv ^ cy;
                cy = sv & cy;
              }
            }
          }
        }
      };

      u32 M0[kQMeta];
      stage_load(0u, M0);

      // ---- base counters of the four groups' paths0 rows: stored planes, or (REC) rebuilt from the recipe ----
      u32 B[L][4];
      auto load_groups = [&](u32 (&P)[LP][4], const u32* planes, u64 unit, int groups) {
        const u32x4* src = (const u32x4*)(planes + unit * 256u) + sub;
#pragma unroll
        for (int j = 0; j < LP / 4; j++) {
#pragma unroll
          for (int d = 0; d < 4; d++) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (j < groups) v = src[j * 64 + 16 * d];
            P[4 * j + 0][d] = v.x; P[4 * j + 1][d] = v.y; P[4 * j + 2][d] = v.z; P[4 * j + 3][d] = v.w;
          }
        }
      };
      if constexpr (!REC) {
        u32 P[LP][4];
        load_groups(P, a.planes0, ((u64)kt * (u64)a.rows0 + (u64)row0) * (u64)a.g0, a.g0);
#pragma unroll
        for (int l = 0; l < L; l++)
#pragma unroll
          for (int d = 0; d < 4; d++) B[l][d] = P[l][d];
      } else {
        const u32 ra = rc0.x, rz = rc0.y & 0x7fffffffu, rinfo = rc0.z, rlov = rc0.w;
        const u32 ro[8] = {rc1.x, rc1.y, rc1.z, rc1.w, rc2.x, rc2.y, rc2.z, rc2.w};
        u32x4 yr[8];
#pragma unroll
        for (int k = 0; k < 8; k++) yr[k] = __builtin_amdgcn_raw_buffer_load_b128(mt, ro[k] + sub16, 0, 0);
        u32 A[LP][4], ZR[LP][4];
        load_groups(A, a.rec_planes_a, ((u64)kt * (u64)a.rec_rows_a + (u64)ra) * (u64)a.rec_ga, a.rec_ga);
        load_groups(ZR, a.rec_planes_z, ((u64)kt * (u64)a.rec_rows_z + (u64)rz) * (u64)a.rec_gz, a.rec_gz);
        const u32 mo = (rinfo & 1u) ? 0xffffffffu : 0u;   // the producing join's list: overlap (A + Z - S) or delta (A + S)
        const u32 rlen = rinfo & kLinfoLenMask;
        u32 S[kQS][4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 r8[8], S4[4];
#pragma unroll
          for (int k = 0; k < 8; k++) r8[k] = yr[k][d];
          sum8(r8, S4);
#pragma unroll
          for (int l = 0; l < kQS; l++) S[l][d] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
        }
        if (__builtin_amdgcn_ballot_w64(rlen > 8u) != 0ull)   // rare: a long list in the recipe
          add_blocks(S, rlen, __builtin_amdgcn_readfirstlane(wave_max_u32(rlen)), a.rec_over, rlov, nullptr);
        // T = X - Y with (X, Y) = (Z, S) for an overlap list, (S, 0) for a delta list; B = A + T
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 bw = 0u, cy = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            const u32 sl = (l < kQS) ? S[l < kQS ? l : 0][d] : 0u;
            const u32 X = mux3(ZR[l][d], sl, mo);
            const u32 Y = sl & mo;
            const u32 T = xor3(X, Y, bw);
            bw = borrow3(X, Y, bw);
            B[l][d] = xor3(A[l][d], T, cy);
            cy = majority(A[l][d], T, cy);
          }
        }
      }

      // ---- first chunk of metadata into LDS, first path's loads on their way ----
      stage_store(M0);
      PathMeta pm;
      u32x4 y[8], zg[GZ];
      fetch(0u, pm, y, zg);

      // ---- one joined path per group and iteration.  The rows are summed first (they leave their registers), then the
      // next path's loads go out, then the counts are put together and tested: one set of row buffers.
      GCRE_QT(tq1);
      GCRE_QT_ADD(0, tq1, tq0);
      for (u32 j = 0; j < npaths; j++) {
        GCRE_QT(ti0);
        // the added row's planes to LDS: lane l holds dword l of GZ plane groups; lane (g, s) reads dwords s + 16 d
#pragma unroll
        for (int jz = 0; jz < GZ; jz++) zst[jz * 64 + lane] = zg[jz];
        const u32 info = pm.info, tot = pm.tot, lh = pm.lh;
        const u32 len = info & kLinfoLenMask;
        const u32 mo = (info & 1u) ? 0xffffffffu : 0u;
        const bool all_overlap = __builtin_amdgcn_ballot_w64((info & 1u) == 0u) == 0ull;
        // 10-20 % of the paths: some group's list is longer than its slot.  The rows of its entries 8..15 (the entries came
        // with the path's other loads) go out before the slot's rows are summed
        const bool any_long = __builtin_amdgcn_ballot_w64(len > 8u) != 0ull;
        u32x4 y2[8];
        if (any_long) {
          const u32 o8[8] = {pm.e2a.x, pm.e2a.y, pm.e2a.z, pm.e2a.w, pm.e2b.x, pm.e2b.y, pm.e2b.z, pm.e2b.w};
#pragma unroll
          for (int k = 0; k < 8; k++) y2[k] = __builtin_amdgcn_raw_buffer_load_b128(mt, o8[k] + sub16, 0, 0);
        }
        const u32 lov = pm.lov;
        u32 S[kQS][4];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 r8[8], S4[4];
#pragma unroll
          for (int k = 0; k < 8; k++) r8[k] = y[k][d];
          sum8(r8, S4);
#pragma unroll
          for (int l = 0; l < kQS; l++) S[l][d] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
        }
        GCRE_QT(ti1);
        GCRE_QT_ADD(1, ti1, ti0);
        if (any_long) add_blocks(S, len, __builtin_amdgcn_readfirstlane(wave_max_u32(len)), a.dover, lov, &y2);
        GCRE_QT(ti2);
        GCRE_QT_ADD(2, ti2, ti1);
        // ---- the next path's loads; at a chunk boundary the next 16 paths' metadata has to be in LDS first (the row
        // buffers are free at this point: the staging registers cost nothing; one exposed round trip per 16 paths)
        if (((j + 1u) & (kQChunk - 1)) == 0u && j + 1u < npaths) {
          u32 Mn[kQMeta];
          stage_load(j + 1u, Mn);
          stage_store(Mn);
        }
        fetch(j + 1u, pm, y, zg);
        GCRE_QT(ti3);
        GCRE_QT_ADD(3, ti3, ti2);

        const u32 lo = lh & 0xffffu, hi = lh >> 16;
        u32 kl[L], kh[L];
#pragma unroll
        for (int l = 0; l < L; l++) {
          kl[l] = (u32)__builtin_amdgcn_sbfe((int)lo, l, 1);
          kh[l] = (u32)__builtin_amdgcn_sbfe((int)hi, l, 1);
        }
        u32 any_m = 0u;
        u32x4 zn[GZ];   // the added row's planes of dwords sub + 16 d, read from LDS one d ahead
#pragma unroll
        for (int jz = 0; jz < GZ; jz++) zn[jz] = zst[jz * 64 + sub];
#pragma unroll
        for (int d = 0; d < 4; d++) {
          u32 Z[LZ];
#pragma unroll
          for (int jz = 0; jz < GZ; jz++) {
            const u32x4 v = zn[jz];
            Z[4 * jz + 0] = v.x; Z[4 * jz + 1] = v.y; Z[4 * jz + 2] = v.z; Z[4 * jz + 3] = v.w;
            if (d < 3) zn[jz] = zst[jz * 64 + sub + 16 * (d + 1)];
          }
          u32 C[L];
          if (all_overlap) {
            // ---- C = B + (Nz - S).  Nz - S >= 0: the overlap is part of the added row ----
            u32 T[LZ];
            u32 bw = 0u;
#pragma unroll
            for (int l = 0; l < LZ; l++) {
              if (l < kQS) {
                T[l] = xor3(Z[l], S[l < kQS ? l : 0][d], bw);
                bw = borrow3(Z[l], S[l < kQS ? l : 0][d], bw);
              } else {
                T[l] = Z[l] ^ bw;
                bw = bw & ~Z[l];
              }
            }
            u32 cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              if (l < LZ) {
                C[l] = xor3(B[l][d], T[l < LZ ? l : 0], cy);
                cy = majority(B[l][d], T[l < LZ ? l : 0], cy);
              } else {
                C[l] = B[l][d] ^ cy;
                cy = B[l][d] & cy;
              }
            }
          } else {
            // ---- some group's list is a delta list (rare): T = X - Y with (X, Y) = (Z, S) or (S, 0) per group ----
            u32 bw = 0u, cy = 0u;
#pragma unroll
            for (int l = 0; l < L; l++) {
              const u32 zl = (l < LZ) ? Z[l < LZ ? l : 0] : 0u;
              const u32 sl = (l < kQS) ? S[l < kQS ? l : 0][d] : 0u;
              const u32 X = mux3(zl, sl, mo);
              const u32 Y = sl & mo;
              const u32 T = xor3(X, Y, bw);
              bw = borrow3(X, Y, bw);
              C[l] = xor3(B[l][d], T, cy);
              cy = majority(B[l][d], T, cy);
            }
          }
          if constexpr (OUT) {
            const u64 rh = (u64)a.out_first + first + j;
            u32x4* dst = (u32x4*)(a.planes_out + (((u64)kt * (u64)a.rows_out + rh) * (u64)a.go) * 256u) + sub + 16 * d;
#pragma unroll
            for (int jo = 0; jo < 4; jo++) {
              if (jo < a.go) {
                u32x4 v = {0u, 0u, 0u, 0u};
                if (4 * jo < L) v = u32x4{C[(4 * jo) % L], (4 * jo + 1 < L) ? C[(4 * jo + 1) % L] : 0u, (4 * jo + 2 < L) ? C[(4 * jo + 2) % L] : 0u,
                                          (4 * jo + 3 < L) ? C[(4 * jo + 3) % L] : 0u};
                dst[jo * 64] = v;
              }
            }
          }
          // ---- interval test: live permutations whose count lies outside [lo, hi] of the path's diagonal ----
          u32 blo = 0u, bhi = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            blo = borrow3(C[l], kl[l], blo);    // C - lo borrows  <=>  C < lo
            bhi = borrow3(kh[l], C[l], bhi);    // hi - C borrows  <=>  C > hi
          }
          u32 m = blo | bhi;
          if (tail_tile) {   // dwords sub + 16 d of the last tile: permutations past K do not exist
            const int lv = (int)k_K - kt * 2048 - (sub + 16 * d) * 32;
            m &= lv >= 32 ? 0xffffffffu : (lv <= 0 ? 0u : ((1u << lv) - 1u));
          }
          if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) continue;
          // ---- the few permutations that can raise a maximum: rebuild each count from the planes, look it up ----
          GCRE_QT(tl0);
          any_m |= m;
          const u32* diag_g = (const u32*)a.t32 + sp_diag_offset(tot);
          u32* nmw = nmax_lds[wave] + (sub + 16 * d);
          while (m != 0u) {
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
              __hip_atomic_fetch_max(nmw + bb[k] * 64, vv[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // ds_max_u32
          }
          dirty = true;
          GCRE_QT(tl1);
          GCRE_QT_ADD(5, tl1, tl0);
        }
        GCRE_QT(ti4);
        GCRE_QT_ADD(4, ti4, ti3);
#ifdef GCRE_IE_TIMING
        tm[7] += 1;
#endif
        if (a.stats) {   // joined paths (of the up to four) with at least one look-up in this tile
          const unsigned long long bal = __builtin_amdgcn_ballot_w64(any_m != 0u);
          if (bal) {
#pragma unroll
            for (u32 g2 = 0; g2 < 4; g2++)
              if (g2 < qcnt && ((bal >> (16 * g2)) & 0xffffull)) n_slow++;
          }
        }
      }
    }
  }
  flush_tile();
#ifdef GCRE_IE_TIMING
  tm[6] = __builtin_amdgcn_s_memtime() - tm_begin;
  if (a.timing && lane == 0)
    for (int i = 0; i < 8; i++) atomicAdd((unsigned long long*)a.timing + i, (unsigned long long)tm[i]);
#endif
  if (a.stats && lane == 0 && n_slow) atomicAdd(a.stats, n_slow);
}

#define GCRE_IEQ_OR(EXPR, LL, GG)                                                  \
  if (out) { if (rec) { EXPR(LL, GG, true, true); } else { EXPR(LL, GG, true, false); } }   \
  else { if (rec) { EXPR(LL, GG, false, true); } else { EXPR(LL, GG, false, false); } }

#ifdef GCRE_IEQ_ONLY   // quick builds while tuning: one variant
#define GCRE_IEQ(EXPR) { const int gz_ = gz; (void)gz_; (void)planes; if (out) { EXPR(10, 2, true, false); } else if (rec) { EXPR(10, 2, false, true); } else { EXPR(10, 2, false, false); } }
#else
#define GCRE_IEQ(EXPR)                                                     \
  if (planes <= 8) { GCRE_IEQ_OR(EXPR, 8, 2) }                             \
  else if (planes <= 10) {                                                 \
    if (gz <= 2) { GCRE_IEQ_OR(EXPR, 10, 2) } else { GCRE_IEQ_OR(EXPR, 10, 3) }        \
  } else if (planes <= 12) {                                               \
    if (gz <= 2) { GCRE_IEQ_OR(EXPR, 12, 2) } else { GCRE_IEQ_OR(EXPR, 12, 3) }        \
  } else {                                                                 \
    if (gz <= 2) { GCRE_IEQ_OR(EXPR, 16, 2) }                              \
    else if (gz == 3) { GCRE_IEQ_OR(EXPR, 16, 3) }                         \
    else { GCRE_IEQ_OR(EXPR, 16, 4) }                                      \
  }
#endif

hipError_t launch_null_ie_quad(const IeArgs& a, int planes, hipStream_t stream) {
  const dim3 grid((unsigned)(8 * a.waves_per_xcd / kIeWaves));
  const dim3 block(64 * kIeWaves);
  const int gz = a.gz;
  const bool out = a.planes_out != nullptr;
  const bool rec = a.rec_slot != nullptr;
#define GCRE_LAUNCHQ(LL, GG, OO, RR) hipLaunchKernelGGL((k_null_ie_q<LL, GG, OO, RR>), grid, block, 0, stream, a)
  GCRE_IEQ(GCRE_LAUNCHQ)
#undef GCRE_LAUNCHQ
  return hipGetLastError();
}

int ieq_max_waves_per_cu(int planes, int gz, bool out, bool rec) {
  int blocks = 0;
  hipError_t e = hipSuccess;
#define GCRE_OCCQ(LL, GG, OO, RR) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_null_ie_q<LL, GG, OO, RR>, 64 * kIeWaves, 0)
  GCRE_IEQ(GCRE_OCCQ)
#undef GCRE_OCCQ
  if (e != hipSuccess || blocks < 1) blocks = 1;
  return blocks * kIeWaves;
}

}  // namespace gcre
