// gcre_ieq.hip -- the pruned method-1 null kernel, "quad" form: one wave works on up to kQSegs segments (two; four in round 2) at a time that
// join the same paths1 rows, and fetches the planes of every added row once for all of them.
//
// Same arithmetic and the same lane layout as k_null_ie_m1 (gcre_ie.hip; lane = 32 permutations of a 2048-permutation
// tile): for every joined path
//
//     count = N0[idx] + Nz[z] - popc(p0[idx] & z & mask_r)        (bit-sliced, reference src/methods.h:73-88)
//
// tested against the pruning ladder's interval [lo, hi] of the path's table diagonal; only what falls outside is looked
// up (methods.h:96-103).  With the bound filter (W = N0 + Nz against [lo + ov, hi], see k_null_ie_m1) nine path-tiles in
// ten never read a mask row, and what is left of k_null_ie_m1's vector-memory traffic is dominated by the two 1-KB plane
// loads of the added row per path-tile: the kernel is bound by the CU's vector-memory pipe at ~20 clocks per wave-load
// (tools/row_gather_rate.hip), not by VALU issue and not by bytes.
//
// All uids with the same pivot gene join the same paths1 rows (the join index gives them the same location and count), so
// their segments walk the same sequence of added rows.  The host groups up to kQSegs such segments into a "quad" (the name stayed from the four-segment form)
// (gcre_host.hip ensure_quads).  The wave keeps the base counters of all of them (kQSegs x L registers) and, for every position
// t of the shared sequence, loads Nz[z_t] once and scores path t of each segment against it: the plane loads per
// path-tile drop from 2 to 0.5, the instruction stream per path-tile is unchanged.
#include "gcre_ie_common.h"

namespace gcre {

// Segments per quad and the occupancy the register allocation aims at.  Round 3 (profiles/r03_quad_shapes.txt): two
// segments at four waves per SIMD beat four at three (25.9 vs 27.4 ms of null kernels per pass on configs[2]: the prologue
// and the exact pass wait on memory, a fourth wave covers more of that than sharing a plane load four ways saves);
// four segments at four waves spill (128 VGPRs), three at four do too.
#ifndef GCRE_REFINE_MAX
#define GCRE_REFINE_MAX 2   // flagged permutations per lane up to which the filter takes its second look (tuning builds)
#endif
#ifndef GCRE_QSEGS
#define GCRE_QSEGS 2
#endif
#ifndef GCRE_QWAVES
#define GCRE_QWAVES 4
#endif
constexpr int kQSegs = GCRE_QSEGS;

// -DGCRE_IE_TIMING: per-section s_memtime sums (a mark only reads the clock)
#ifdef GCRE_IE_TIMING
#define GCRE_QT(var) const u64 var = __builtin_amdgcn_s_memtime()
#define GCRE_QT_ADD(i, t1, t0) tm[i] += (t1) - (t0)
#else
#define GCRE_QT(var)
#define GCRE_QT_ADD(i, t1, t0)
#endif

template <int L, int GZ, bool REC>
__global__ __launch_bounds__(64 * kIeWaves) __attribute__((amdgpu_waves_per_eu(GCRE_QWAVES))) void k_null_ie_q(const IeArgs a) {
  constexpr int LP = (L + 3) / 4 * 4;
  constexpr int LZ = 4 * GZ;
  static_assert(L >= 8 && L <= 16 && GZ >= 2 && LZ <= LP, "planes come in groups of 4");
  typedef u32 __attribute__((ext_vector_type(8))) u32x8;
  __shared__ u32 nmax_lds[kIeWaves][32 * 64];   // the waves' running maxima [bit][lane]
  __shared__ u32 wq_state[kIeWaves][8];
  // look-ups are queued (cell of the table, slot of the maximum) and made 64 at a time: one memory round trip per 64
  // look-ups instead of one per path (as k_null_ie_m2; a threshold read before a drain is only lower, never wrong)
  constexpr u32 kLqCap = 128u;
  __shared__ u32 lq_lds[kIeWaves][2][kLqCap];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const u32 lane4 = (u32)lane * 4u;
  u32* nm = nmax_lds[wave] + lane;
#pragma unroll
  for (int q = 0; q < 32; q++) nm[q * 64] = 0u;
  u32 (*lq)[kLqCap] = lq_lds[wave];
  u32 lq_n = 0u;   // entries waiting (wave-uniform)
  auto lq_drain = [&]() {
    for (u32 base = 0u; base < lq_n; base += 64u) {
      const u32 i = base + (u32)lane;
      if (i < lq_n)
        __hip_atomic_fetch_max(nmax_lds[wave] + lq[1][i], ((const u32*)a.t32)[lq[0][i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);   // ds_max_u32
    }
    lq_n = 0u;
  };
  // Launch constants that steer control flow or end up in buffer descriptors, pinned to scalar registers: the compiler
  // re-loads kernel arguments on both sides of the `lane == 0` branch around the ticket atomic, and whatever is derived
  // from such a pair counts as divergent (waterfall loops around buffer loads, vector-register loop counters).
  auto uni = [](u32 v) -> u32 { return (u32)__builtin_amdgcn_readfirstlane((int)v); };
  auto uni_ptr = [&](const void* p) -> const char* { return (const char*)(((u64)uni((u32)((u64)p >> 32)) << 32) | (u64)uni((u32)(u64)p)); };
  const u32 k_quad_begin = uni((u32)a.quad_begin), k_quad_end = uni((u32)a.quad_end), k_batch = uni((u32)a.batch);
  const u32 k_nkt = uni((u32)a.nkt), k_K = uni((u32)a.K), k_mt_rows = uni(a.mt_rows), k_lstride = uni((u32)a.ladder_stride);
  const u32 k_lad_mode = uni((u32)a.lad_mode), k_score_segs = uni(a.score_segs);
  const char* k_mt = uni_ptr(a.mt);
  const char* k_planesz = uni_ptr(a.planesz);

  int cur_kt = -1;
  u32 valid = 0u;
  u32 lad_base = (k_lad_mode == 0u) ? 0u : ((u32)kLadderLevels - 1u + k_lad_mode) * k_lstride;
  const u32 lad_keep = (u32)kLadderLevels * k_lstride;
  u32 n_slow = 0u;
#ifdef GCRE_IE_TIMING
  u64 tm[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // header + loads issued, base counters, filter intervals, filter pass, exact pass, exchange, total, quads
  const u64 tm_begin = __builtin_amdgcn_s_memtime();
#endif
  __amdgpu_buffer_rsrc_t mt = __builtin_amdgcn_make_buffer_rsrc((void*)k_mt, 0, 0x7fffffff, 0x00020000);

  // publish the wave's maxima, read everybody's, set the threshold level to the smallest running maximum of the tile's
  // live permutations (a stale read only lowers it: still exact)
  // publish the wave's maxima, read everybody's, set the threshold level to the smallest running maximum of the tile's
  // live permutations (a stale read only lowers it: still exact).  The merged values stay in LDS: nm[] is then the best
  // maximum this wave KNOWS of every permutation of the tile (its own finds + the others' as of the last exchange), which
  // is what the per-permutation second look of the filter (refine, below) tests against.
  auto merge_global = [&](bool want_theta) {
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
      if (own > g) atomicMax(out + q, own);     // only what this wave raised itself can be above the global value
      const u32 v = own > g ? own : g;
      nm[q * 64] = want_theta ? v : 0u;         // the tile ends: the next tile starts from nothing
      if ((valid >> q) & 1u) lo = v < lo ? v : lo;
    }
    return lo;
  };
  auto exchange = [&]() {
    lq_drain();
    u32 theta = __builtin_amdgcn_readfirstlane(wave_min_u32(merge_global(true)));
    if (theta == 0xffffffffu) theta = 0u;
    int j = (int)(__uint_as_float(theta) * (float)kLadderPerUnit);
    j = j < 0 ? 0 : (j > kLadderLevels - 1 ? kLadderLevels - 1 : j);
    lad_base = (u32)j * k_lstride;
  };
  auto flush_tile = [&]() {
    lq_drain();
    if (cur_kt >= 0) (void)merge_global(false);
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
      mt = __builtin_amdgcn_make_buffer_rsrc((void*)(k_mt + (size_t)kt * k_mt_rows * 256u), 0, 0x7fffffff, 0x00020000);
      const int live = (int)k_K - kt * 2048 - lane * 32;
      valid = live >= 32 ? 0xffffffffu : (live <= 0 ? 0u : ((1u << live) - 1u));
      if (k_lad_mode == 0u) lad_base = 0u;
      since = 0;
      period = 1;
    }
    // The table entries of a quad's segments -- (row0, first, n) and, with a recipe, the twelve words k_fill_rec_segs
    // gathered per segment -- are wave-uniform, but a scalar load of them misses all the way to HBM (the tables are tens
    // of MB and every entry is read once): four dependent round trips per quad.  They are fetched one quad ahead into
    // one vector register per segment (lane l = word l) and moved to scalar registers with v_readlane.
    auto load_hdr = [&](u32 qe_, u32 (&h)[kQSegs]) {
      const u32 cnt_ = (qe_ >> 30) + 1u, s0_ = qe_ & 0x3fffffffu;
#pragma unroll
      for (int g = 0; g < kQSegs; g++) {
        h[g] = 0u;
        if ((u32)g < cnt_) {
          const u32* src = (const u32*)a.segs + (u64)(s0_ + (u32)g) * 3u + (u32)(lane < 3 ? lane : 0);
          if constexpr (REC) {
            if (lane >= 4 && lane < 16) src = a.rec_segs + (u64)(s0_ + (u32)g) * kRecSegWords + (u32)(lane - 4);
          }
          h[g] = *src;
        }
      }
    };
    u32 hdr_n[kQSegs] = {};
    u32 qe_n = 0u;
    bool have_next = false;
    for (u32 qi = q_lo; qi < q_hi; qi++) {
      // ---- the quad: up to kQSegs consecutive segments of the table, same added rows, same length ----
      GCRE_QT(t0);
      u32 hdr[kQSegs];
      u32 qe;
      if (have_next) {
        qe = qe_n;
#pragma unroll
        for (int g = 0; g < kQSegs; g++) hdr[g] = hdr_n[g];
      } else {
        qe = uni(((const u32 GCRE_CONSTANT*)a.quads)[qi]);
        load_hdr(qe, hdr);
      }
      have_next = qi + 1u < q_hi;
      if (have_next) qe_n = uni(((const u32 GCRE_CONSTANT*)a.quads)[qi + 1u]);   // its headers go out below, behind this quad's loads
      const u32 qcnt = (qe >> 30) + 1u;
      const u32 s0 = qe & 0x3fffffffu;
      const u32 npaths = rdlane(hdr[0], 2);
      const u32 last = npaths - 1u;
      GCRE_QT(te0);
      if (k_lad_mode == 0u && ++since >= period) {
        exchange();
        since = 0;
        period = period < kIeRefresh ? period * 2 : kIeRefresh;
      }
      GCRE_QT(te1);
      GCRE_QT_ADD(5, te1, te0);
      const u32 lad_row = (s0 < k_score_segs) ? lad_base : lad_keep;

      // ---- per segment: the metadata of its paths (lane t <-> path first + t) and its base counters ----
      u32 infov[kQSegs], lhv[kQSegs], lfv[kQSegs], firstg[kQSegs];
      u32 B[kQSegs][L];
      u32 zunit = 0u;
      auto load_groups = [&](u32 (&P)[LP], const u32* planes, u64 unit, int groups) {
        __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc((void*)(uni_ptr(planes) + unit * 1024u), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < LP / 4; j++) {
          u32x4 v = {0u, 0u, 0u, 0u};
          if (j < groups) v = __builtin_amdgcn_raw_buffer_load_b128(rp, lane4 * 4u + (u32)j * 1024u, 0, 0);
          P[4 * j + 0] = v.x; P[4 * j + 1] = v.y; P[4 * j + 2] = v.z; P[4 * j + 3] = v.w;
        }
      };
      // (a) everything the quad's segments need from memory goes out together: their paths' metadata, the planes their
      // base counters start from and -- with a recipe -- the mask rows of the producing join's list.  One round trip per
      // quad, not one per segment.
      u32 totv[kQSegs];
      u32 yr[REC ? kQSegs : 1][8];
      u32 rinfo_g[kQSegs] = {}, rlov_g[kQSegs] = {}, rz_g[kQSegs] = {};
      // the paths' metadata first: the ladder gather below needs the carrier totals and must not queue behind the planes
#pragma unroll
      for (int g = 0; g < kQSegs; g++) {
        totv[g] = 0u;
        if ((u32)g < qcnt) {
          const u32 first = rdlane(hdr[g], 1);
          firstg[g] = first;
          const u32 qv = first + (((u32)lane < npaths) ? (u32)lane : 0u);
          infov[g] = a.linfo[qv];
          totv[g] = a.tot[qv];
          if (g == 0) zunit = ((u32)kt * (u32)a.rowsz + (a.rowz[qv] & 0x7fffffffu)) * (u32)a.gz;   // the same for every segment of the quad
        }
      }
#pragma unroll
      for (int g = 0; g < kQSegs; g++) {
        if ((u32)g < qcnt) {
          const u32 row0 = rdlane(hdr[g], 0);
          u32 Bp[LP];
          if constexpr (!REC) {
            load_groups(Bp, a.planes0, ((u64)kt * (u64)a.rows0 + (u64)row0) * (u64)a.g0, a.g0);
          } else {
            // the segment's recipe words sit next to the segment table (k_fill_rec_segs): paths0 row of the producing
            // join, row it added, list info, where a long list continues, the list's first 8 entries
            const u32 ra = rdlane(hdr[g], 4);
            rz_g[g] = rdlane(hdr[g], 5) & 0x7fffffffu;
            rinfo_g[g] = rdlane(hdr[g], 6);
            rlov_g[g] = rdlane(hdr[g], 7);
            u32 ro[8];
#pragma unroll
            for (int j = 0; j < 8; j++) ro[j] = rdlane(hdr[g], 8 + j);
            const u32 rtrue = (rinfo_g[g] & kLinfoLenMask) - (rinfo_g[g] >> 28);   // entries that are not padding
#pragma unroll
            for (int j = 0; j < 8; j++) {
              yr[REC ? g : 0][j] = 0u;
              if ((u32)j < rtrue) yr[REC ? g : 0][j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, ro[j], 0);
            }
            // the row's carrier total bounds the counts of everything it was made from: planes above it are zero and are
            // not read (3 KB -> 2 KB of HBM per segment and tile for most rows: this load is the kernel's HBM traffic)
            const int ga_need = (int)((rinfo_g[g] >> 1) & 3u) + 1;
            load_groups(Bp, a.rec_planes_a, ((u64)kt * (u64)a.rec_rows_a + (u64)ra) * (u64)a.rec_ga, ga_need < a.rec_ga ? ga_need : a.rec_ga);
          }
#pragma unroll
          for (int l = 0; l < L; l++) B[g][l] = Bp[l];
        }
      }
      if (have_next) load_hdr(qe_n, hdr_n);
      GCRE_QT(t1);
      GCRE_QT_ADD(0, t1, t0);
      // (b) with a recipe: row = A + Z -/+ list.  The segments of a quad end in the same pivot gene, so the row the
      // producing join added -- that gene -- is the same for all of them: its planes are loaded once
      if constexpr (REC) {
        u32 ZR[LP];
        load_groups(ZR, a.rec_planes_z, ((u64)kt * (u64)a.rec_rows_z + (u64)rz_g[0]) * (u64)a.rec_gz, a.rec_gz);
#pragma unroll
        for (int g = 0; g < kQSegs; g++) {
          if ((u32)g < qcnt) {
            if (rz_g[g] != rz_g[0]) load_groups(ZR, a.rec_planes_z, ((u64)kt * (u64)a.rec_rows_z + (u64)rz_g[g]) * (u64)a.rec_gz, a.rec_gz);
            const u32 rinfo = rinfo_g[g];
            const u32 rlen = rinfo & kLinfoLenMask;
            u32 S[L];
            {
              u32 S4[4];
              sum8(yr[REC ? g : 0], S4);
#pragma unroll
              for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
            }
            if (rlen > 8u) {   // the producing join's list was long: the rest of it, 8 entries at a time
              const u32 GCRE_CONSTANT* more = (const u32 GCRE_CONSTANT*)(a.rec_over + rlov_g[g]);
              for (u32 p = 0u; p + 8u < rlen; p += 8u) {
                const u32x8 o8 = *(const u32x8 GCRE_CONSTANT*)(more + p);
                u32 yy[8], s4[4];
#pragma unroll
                for (int j = 0; j < 8; j++) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, uni(o8[j]), 0);
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
                const u32 s1_ = xor3(B[g][l], ZR[l], cy);
                cy = majority(B[g][l], ZR[l], cy);
                B[g][l] = xor3(s1_, S[l], bw);
                bw = borrow3(s1_, S[l], bw);
              }
            } else {            // B = A + S
              u32 cy = 0u;
#pragma unroll
              for (int l = 0; l < L; l++) {
                const u32 bl = B[g][l];
                B[g][l] = xor3(bl, S[l], cy);
                cy = majority(bl, S[l], cy);
              }
            }
            if (rz_g[g] != rz_g[0]) load_groups(ZR, a.rec_planes_z, ((u64)kt * (u64)a.rec_rows_z + (u64)rz_g[0]) * (u64)a.rec_gz, a.rec_gz);
          }
        }
      }
      GCRE_QT(t2);
      GCRE_QT_ADD(1, t2, t1);
      // (c) the filter's interval per path, ready to use: [lo + ov, hi] packed as hi << 16 | lo; the empty interval (lo 1,
      // hi 0: "everything is outside") for delta lists and for bounds the margin pushes out of range
#pragma unroll
      for (int g = 0; g < kQSegs; g++) {
        if ((u32)g < qcnt) {
          const u32 lh = a.ladder[lad_row + totv[g]];
          const u32 r0 = infov[g];
          const u32 ov = (r0 & kLinfoLenMask) - (r0 >> 28);
          const u32 lo = lh & 0xffffu;
          const u32 lo2 = lo ? lo + ov : 0u;              // counts are never negative: lo = 0 needs no margin
          lhv[g] = lh;
          lfv[g] = ((r0 & 1u) == 0u || (lo2 >> L) != 0u) ? 1u : ((lh & 0xffff0000u) | lo2);
        }
      }

      GCRE_QT(t3);
      GCRE_QT_ADD(2, t3, t2);
      const u32x8 GCRE_CONSTANT* slots = (const u32x8 GCRE_CONSTANT*)a.dlist;
      auto issue = [&](u32 t2, u32 (&ZZ)[LZ]) {
        __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc((void*)(k_planesz + (u64)rdlane(zunit, t2) * 1024u), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int j = 0; j < GZ; j++) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rz, lane4 * 4u + (u32)j * 1024u, 0, 0);
          ZZ[4 * j + 0] = v.x; ZZ[4 * j + 1] = v.y; ZZ[4 * j + 2] = v.z; ZZ[4 * j + 3] = v.w;
        }
      };
      auto at = [&](u32 t2) -> u32 { return t2 < last ? t2 : last; };

      // ---- first pass: the bound filter.  Path t of segment g against the planes Z of the row the quad's segments all
      // add: W = B_g + Z is inside [lo + ov, hi] for every live permutation <=> no count of the path can raise a maximum
      // whatever its overlap rows say.  Paths that fail (about one in ten) -- and the rare delta-list paths -- are only
      // marked, one bit per path in a scalar mask per segment: nothing in this loop waits for anything but the planes.
      u64 todo[kQSegs] = {};
      auto filter_f = [&](int g, u32 t, const u32 (&Bg)[L], const u32 (&Z)[LZ]) {
        // rows of another shard (the ladder's all-inside row): nothing of theirs is scored, so nothing is examined -- neither
        // a carry out of the top plane nor a delta list may send such a path to the second look or the exact pass
        if (lad_row == lad_keep) return;
        const u32 lf = rdlane(lfv[g], t);   // hi << 16 | lo + ov
        u32 cy = 0u, blo = 0u, bhi = 0u;
#pragma unroll
        for (int l = 0; l < L; l++) {
          u32 w;
          if (l < LZ) {
            w = xor3(Bg[l], Z[l < LZ ? l : 0], cy);
            cy = majority(Bg[l], Z[l < LZ ? l : 0], cy);
          } else {
            w = Bg[l] ^ cy;
            cy = Bg[l] & cy;
          }
          const u32 kl = (u32)__builtin_amdgcn_sbfe((int)lf, l, 1);
          const u32 kh = (u32)__builtin_amdgcn_sbfe((int)lf, 16 + l, 1);
          blo = borrow3(w, kl, blo);    // W < lo + ov
          bhi = borrow3(kh, w, bhi);    // W > hi
        }
        // (W counts the overlap twice and can pass 2^L although no count of the path does: a carry out of the top plane is
        // "above hi", not a small W)
        u32 fm = (blo | bhi | cy) & valid;
        if (__builtin_amdgcn_ballot_w64(fm != 0u) == 0ull) return;
        // ---- a second look, permutation by permutation.  The interval above belongs to the LOWEST running maximum of the
        // tile's 2048 permutations; a flagged permutation only matters if its count can leave the (wider) interval of its
        // OWN maximum as far as this wave knows it (nm[], refreshed at every exchange).  count = W - S with 0 <= S <= ov, so
        // [W - ov, W] inside that interval settles it without a mask row: most flagged paths end here instead of in the
        // exact pass (a path costs ~300 instructions there).  Overlap lists only, thresholds from the maxima only. ----
        const u32 info = rdlane(infov[g], t);
        // (a path with many flagged permutations -- dense genotypes, where the margin eats the interval -- is not worth
        // the look: it goes to the exact pass as before)
        if (k_lad_mode == 0u && (info & 1u) != 0u && __builtin_amdgcn_ballot_w64(__builtin_popcount(fm) > GCRE_REFINE_MAX) == 0ull) {
          const u32 ov = (info & kLinfoLenMask) - (info >> 28);
          const u32* lad_t = a.ladder + rdlane(totv[g], t);
          u32 Wp[L];
          u32 c2 = 0u;
          {
#pragma unroll
            for (int l = 0; l < L; l++) {
              if (l < LZ) {
                Wp[l] = xor3(Bg[l], Z[l < LZ ? l : 0], c2);
                c2 = majority(Bg[l], Z[l < LZ ? l : 0], c2);
              } else {
                Wp[l] = Bg[l] ^ c2;
                c2 = Bg[l] & c2;
              }
            }
          }
          bool unsafe = false;
          while (__builtin_amdgcn_ballot_w64(fm != 0u) != 0ull) {   // one flagged permutation per lane and round
            const bool has = fm != 0u;
            const u32 bb = has ? (u32)__builtin_ctz(fm) : 0u;
            fm &= fm - 1u;
            u32 cnt = ((c2 >> bb) & 1u) << L;   // the carry out of the top plane is part of W
#pragma unroll
            for (int l = 0; l < L; l++) cnt |= ((Wp[l] >> bb) & 1u) << l;
            if (has) {
              int j = (int)(__uint_as_float(nm[bb * 64]) * (float)kLadderPerUnit);
              j = j < 0 ? 0 : (j > kLadderLevels - 1 ? kLadderLevels - 1 : j);
              const u32 lh = lad_t[(u32)j * k_lstride];
              const u32 lo = lh & 0xffffu, hi = lh >> 16;
              // every count in [max(0, W - ov), W] inside [lo, hi]?  (counts are never negative: lo = 0 needs no margin)
              if (cnt > hi || (lo != 0u && cnt < lo + ov)) unsafe = true;
            }
          }
          if (__builtin_amdgcn_ballot_w64(unsafe) == 0ull) return;
        }
        todo[g] |= 1ull << t;
      };

      // the planes of the next added row are in flight while up to four paths are tested against the current one
      {
        u32 ZA[LZ], ZB[LZ];
        issue(0u, ZA);
        for (u32 t = 0; t < npaths; t += 2) {
          issue(at(t + 1), ZB);
#pragma unroll
          for (int g = 0; g < kQSegs; g++)
            if ((u32)g < qcnt) filter_f(g, t, B[g], ZA);
          if (t + 1 < npaths) {
            issue(at(t + 2), ZA);
#pragma unroll
            for (int g = 0; g < kQSegs; g++)
              if ((u32)g < qcnt) filter_f(g, t + 1, B[g], ZB);
          }
        }
      }

      GCRE_QT(t4);
      GCRE_QT_ADD(3, t4, t3);
      // ---- second pass: the marked paths, exactly.  The planes of the added row again, the mask rows of the list's real
      // entries, count = B + Nz - S (or B + S), the interval test, look-ups for what falls outside -- with the loads of
      // the next marked path in flight while this one is computed.
      struct Item { u32 t, info, lov, tot, lh; u32x8 o; };
      auto item_meta = [&](int g, u32 t) -> Item {
        Item it;
        const u64 q = (u64)firstg[g] + t;
        it.t = t;
        it.info = rdlane(infov[g], t);
        it.lh = rdlane(lhv[g], t);
        it.lov = (u32)q;                  // where a long list continues is only read for a long list (exact_f)
        it.tot = rdlane(totv[g], t);
        it.o = slots[q];
        return it;
      };
      auto item_issue = [&](const Item& it, u32 (&yy)[8], u32 (&ZZ)[LZ]) {
        const u32 ov = (it.info & kLinfoLenMask) - (it.info >> 28);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          yy[j] = 0u;
          if ((u32)j < ov) yy[j] = __builtin_amdgcn_raw_buffer_load_b32(mt, lane4, it.o[j], 0);   // padding entries are not fetched
        }
        if (it.info & 1u) issue(it.t, ZZ);
      };
      auto exact_f = [&](const Item& it, const u32 (&Bg)[L], const u32 (&y)[8], const u32 (&Z)[LZ]) {
        const u32 len = it.info & kLinfoLenMask;
        const bool overlap = (it.info & 1u) != 0u;
        const u32 lo = it.lh & 0xffffu, hi = it.lh >> 16;
        u32 S4[4];
        sum8(y, S4);
        u32 S[L];
#pragma unroll
        for (int l = 0; l < L; l++) S[l] = (l < 4) ? S4[l < 4 ? l : 0] : 0u;
        if (len > 8u) {   // long list: further blocks of 8 entries
          const u32 GCRE_CONSTANT* more = (const u32 GCRE_CONSTANT*)(a.dover + ((const u32 GCRE_CONSTANT*)a.lover)[it.lov]);
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
        if (overlap) {   // C = B + Nz - S
          u32 cy = 0u, bw = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            const u32 zl = (l < LZ) ? Z[l < LZ ? l : 0] : 0u;
            const u32 w = xor3(Bg[l], zl, cy);
            cy = majority(Bg[l], zl, cy);
            C[l] = xor3(w, S[l], bw);
            bw = borrow3(w, S[l], bw);
          }
        } else {         // C = B + S
          u32 cy = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) {
            C[l] = xor3(Bg[l], S[l], cy);
            cy = majority(Bg[l], S[l], cy);
          }
        }
        u32 blo = 0u, bhi = 0u;
#pragma unroll
        for (int l = 0; l < L; l++) {
          const u32 kl = (u32)__builtin_amdgcn_sbfe((int)lo, l, 1);
          const u32 kh = (u32)__builtin_amdgcn_sbfe((int)hi, l, 1);
          blo = borrow3(C[l], kl, blo);    // C < lo
          bhi = borrow3(kh, C[l], bhi);    // C > hi
        }
        u32 m = (blo | bhi) & valid;
        if (__builtin_amdgcn_ballot_w64(m != 0u) == 0ull) return;
        n_slow++;
        const u32 diag = sp_diag_offset(it.tot);
        while (__builtin_amdgcn_ballot_w64(m != 0u) != 0ull) {   // one permutation per lane and round
          if (lq_n + 64u > kLqCap) lq_drain();
          const bool has = m != 0u;
          const u32 bb = has ? (u32)__builtin_ctz(m) : 0u;
          m &= m - 1u;
          u32 cnt = 0u;
#pragma unroll
          for (int l = 0; l < L; l++) cnt |= ((C[l] >> bb) & 1u) << l;
          const u64 hm = __builtin_amdgcn_ballot_w64(has);
          const u32 pos = lq_n + __builtin_amdgcn_mbcnt_hi((u32)(hm >> 32), __builtin_amdgcn_mbcnt_lo((u32)hm, 0u));
          if (has) {
            lq[0][pos] = diag + cnt;
            lq[1][pos] = bb * 64u + (u32)lane;
          }
          lq_n += (u32)__builtin_popcountll(hm);
        }
      };
#pragma unroll
      for (int g = 0; g < kQSegs; g++) {
        if ((u32)g < qcnt && todo[g] != 0ull) {
          u64 m = todo[g];
          u32 yA[8], yB[8], ZA[LZ], ZB[LZ];
#pragma unroll
          for (int l = 0; l < LZ; l++) ZA[l] = ZB[l] = 0u;
          // three stages: the list entries and totals of a marked path are read (scalar loads) while the path before it
          // has its rows and planes in flight and the one before that is computed
          auto pop = [&]() -> u32 {
            const u32 t = (u32)__builtin_ctzll(m);
            m &= m - 1ull;
            return t;
          };
          Item cur = item_meta(g, pop()), nxt = cur, aft = cur;
          bool has_n = m != 0ull, has_a = false;
          if (has_n) nxt = item_meta(g, pop());
          item_issue(cur, yA, ZA);
          for (;;) {
            if (has_n) item_issue(nxt, yB, ZB);
            has_a = m != 0ull;
            if (has_a) aft = item_meta(g, pop());
            exact_f(cur, B[g], yA, ZA);
            if (!has_n) break;
            cur = nxt; nxt = aft; has_n = has_a;
            if (has_n) item_issue(nxt, yA, ZA);
            has_a = m != 0ull;
            if (has_a) aft = item_meta(g, pop());
            exact_f(cur, B[g], yB, ZB);
            if (!has_n) break;
            cur = nxt; nxt = aft; has_n = has_a;
          }
        }
      }
      GCRE_QT(t5);
      GCRE_QT_ADD(4, t5, t4);
#ifdef GCRE_IE_TIMING
      tm[7] += 1;
#endif
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

#define GCRE_IEQ_R(EXPR, LL, GG) if (rec) { EXPR(LL, GG, true); } else { EXPR(LL, GG, false); }

#ifdef GCRE_IEQ_ONLY   // quick builds while tuning: one variant
#define GCRE_IEQ(EXPR) { (void)gz; (void)planes; GCRE_IEQ_R(EXPR, 10, 2) }
#else
#define GCRE_IEQ(EXPR)                                                     \
  if (planes <= 8) { GCRE_IEQ_R(EXPR, 8, 2) }                              \
  else if (planes <= 10) {                                                 \
    if (gz <= 2) { GCRE_IEQ_R(EXPR, 10, 2) } else { GCRE_IEQ_R(EXPR, 10, 3) }        \
  } else if (planes <= 11) {                                               \
    if (gz <= 2) { GCRE_IEQ_R(EXPR, 11, 2) } else { GCRE_IEQ_R(EXPR, 11, 3) }        \
  } else if (planes <= 12) {                                               \
    if (gz <= 2) { GCRE_IEQ_R(EXPR, 12, 2) } else { GCRE_IEQ_R(EXPR, 12, 3) }        \
  } else {                                                                 \
    if (gz <= 2) { GCRE_IEQ_R(EXPR, 16, 2) }                               \
    else if (gz == 3) { GCRE_IEQ_R(EXPR, 16, 3) }                          \
    else { GCRE_IEQ_R(EXPR, 16, 4) }                                       \
  }
#endif

// quad form: method 1, pruned, no plane output (kept joins stay on k_null_ie_m1)
hipError_t launch_null_ie_quad(const IeArgs& a, int planes, hipStream_t stream) {
  const dim3 grid((unsigned)(8 * a.waves_per_xcd / kIeWaves));
  const dim3 block(64 * kIeWaves);
  const int gz = a.gz;
  const bool rec = a.rec_slot != nullptr;
#define GCRE_LAUNCHQ(LL, GG, RR) hipLaunchKernelGGL((k_null_ie_q<LL, GG, RR>), grid, block, 0, stream, a)
  GCRE_IEQ(GCRE_LAUNCHQ)
#undef GCRE_LAUNCHQ
  return hipGetLastError();
}

int ieq_quad_segs() { return kQSegs; }

int ieq_max_waves_per_cu(int planes, int gz, bool rec) {
  int blocks = 0;
  hipError_t e = hipSuccess;
#define GCRE_OCCQ(LL, GG, RR) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks, k_null_ie_q<LL, GG, RR>, 64 * kIeWaves, 0)
  GCRE_IEQ(GCRE_OCCQ)
#undef GCRE_OCCQ
  if (e != hipSuccess || blocks < 1) blocks = 1;
  return blocks * kIeWaves;
}

}  // namespace gcre
