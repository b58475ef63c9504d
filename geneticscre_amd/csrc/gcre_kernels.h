// gcre_kernels.h -- launch interface between the host library (gcre_host.hip) and the gfx950 kernels
// (gcre_kernels.hip).  Internal; the public boundary is include/gcre_hip.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gcre {

// Device layout (DESIGN.md "Data layout in HBM"):
//   path rows : uint64 [rows][S], S = M * Wp, Wp = ceil(n/64) rounded up to 4 words (32-byte chunks);
//               method 2 keeps the (+) half in words [0,Wp) and the (-) half in [Wp,2Wp)
//   masks     : uint32 [W32p][Kpad], W32p = 2*Wp dwords, word-major / permutation-minor
//               (the reference's layout, src/join_base.cpp:109, at 32-bit granularity), Kpad % 512 == 0
//   tables    : "diagonal-major": entry [t][i] = table[i][t-i] at index t*(t+1)/2 + i, t = carriers on the
//               path -- one path only ever touches one diagonal (src/methods.h:96-100)
struct Geometry {
  int method;   // M
  int n;        // patients
  int n_cases;
  int W;        // ceil(n/64)
  int Wp;       // padded words per half
  int S;        // row stride in uint64
  int K;        // iterations requested
  int Kpad;
  int TD;       // number of table diagonals = 64*Wp + 1
};

constexpr int kPermTileMax = 2048;     // Kpad granularity (the sparse kernel's permutation tile)
constexpr int kNullBlock = 256;        // threads per block of the null kernel

struct NullArgs {
  const uint32_t* p0;       // paths0 rows (dword view)
  const uint32_t* p1;       // paths1 rows
  const uint32_t* masks;    // [W32p][Kpad]
  const uint32_t* row0;     // per joined path in this launch: row of paths0
  const uint32_t* row1;     // row of paths1, bit 31 = swap (+)/(-) halves of path1 (method 2)
  const uint32_t* tot;      // M entries per path: carriers (M=1) or carriers in (+),(-) halves (M=2)
  const float* t32;         // M=1: sanitised f32 null table, diagonal-major
  const double* d64;        // M=2: f64 vtmax, diagonal-major
  uint32_t* null_bits;      // [Kpad] running maxima as u32 bit patterns of non-negative floats
  int64_t npaths;           // joined paths in this launch
  int64_t npt;              // path tiles
  int S32;                  // row stride in dwords
  int W32p;                 // dwords per half
  int Kpad;
  int nkt;                  // permutation tiles
  int pgroups;              // blocks per permutation tile
};

struct NullConfig {
  int R;        // permutations per lane
  int TPW;      // joined paths per wave per tile
  int perm_tile;   // 64 * R
  int path_tile;   // 4 * TPW
};

NullConfig null_config(int method, int K);
// grid is derived from the args (nkt * pgroups blocks)
hipError_t launch_null(const NullArgs& a, int method, const NullConfig& cfg, hipStream_t stream);

// ---- sparse / bit-sliced null kernel (gcre_sparse.hip) ----
struct SparseSeg {
  uint32_t row0;    // row of paths0 shared by the segment's joined paths
  uint32_t first;   // first joined path of the segment, relative to the launch
  uint32_t n;       // joined paths in the segment
};

struct SparseArgs {
  const uint32_t* mt;        // transposed masks [nkt][mt_rows][64]; row mt_rows-1 is all zero
  const uint32_t* tot;       // carriers per joined path of the launch
  const SparseSeg* segs;
  const uint64_t* loff0;     // CSR bit lists of paths0: offsets [rows+1] ...
  const uint32_t* lidx0;     // ... and entries = patient << 8 (byte offset of the patient's mask row), 4-padded
  const uint64_t* doff;      // per joined path of the launch: [count+1] offsets into dlist
  const uint32_t* dlist;     // entries of the bits paths1 adds on top of paths0, same encoding, 4-padded
  const float* t32;          // method 1 null table
  const double* d64;         // method 2 null table (vtmax)
  uint32_t* null_bits;
  int64_t nsegs;
  int nkt;                   // 2048-permutation tiles
  int waves_per_xcd;         // persistent waves per XCD (grid = 8 * waves_per_xcd / 4 blocks)
  uint32_t mt_rows;
  uint32_t zoff;             // byte offset of the all-zero mask row = (mt_rows - 1) * 256
  int ablate;                // diagnostics only (GCRE_SPARSE_ABLATE)
};
constexpr int kSparseTile = 2048;
constexpr int kSparseSegMax = 64;
hipError_t launch_null_sparse(const SparseArgs& a, int method, int planes, hipStream_t stream);
int sparse_max_waves_per_cu(int method, int planes);   // resident waves per CU of the variant chosen for `planes` counter planes
// ---- inclusion-exclusion null kernel on count planes (gcre_ie.hip) ----
constexpr int kRecSegWords = 12;
// linfo word of a list: padded length (multiple of 8, >= 8) | mode (bit 0: 1 = overlap list) | (padding entries) << 28,
// so that the list's true length = padded length - padding is known without walking it (the bound filter of k_null_ie_m1)
constexpr uint32_t kLinfoLenMask = 0x0ffffff8u;
constexpr int kLadderLevels = 256;   // pruning thresholds j / kLadderPerUnit, j = 0 .. kLadderLevels-1
constexpr int kLadderPerUnit = 8;
constexpr int kLadder2Levels = 352;  // signed method: rows r <-> threshold r / (2 kLadderPerUnit), up to 4/3 of the method-1 range
struct IeArgs {
  const uint32_t* mt;        // transposed masks, as SparseArgs
  const uint32_t* tot;       // carriers per joined path (and half) of the launch
  const uint32_t* rowz;      // per joined path: row of the reduced operand | swap << 31
  const SparseSeg* segs;
  const uint32_t* planes0;   // count planes of paths0 [tile][row*M+h][g0][64][4], or nullptr: stream loff0/lidx0
  const uint32_t* planesz;   // count planes of the reduced operand [tile][row*M+h][gz][64][4] (mode-1 paths)
  uint32_t rows0, rowsz, rows_out;   // row-halves (rows * M) of the three plane arrays
  // paths0 without stored planes but with the recipe of the join that produced it (rec_slot != nullptr): row r of
  // paths0 = row rec_row0[r] of set A | row rec_rowz[r] of set Z (bit 31: halves swapped), list info / slot / overflow per
  // list (row * M + half) as that join's inspector left them; rec_rows_a / rec_rows_z = row-halves of A / Z
  const uint32_t* rec_row0;
  const uint32_t* rec_rowz;
  const uint32_t* rec_linfo;
  const uint32_t* rec_lover;
  const uint32_t* rec_slot;
  const uint32_t* rec_over;
  const uint32_t* rec_planes_a;
  const uint32_t* rec_planes_z;
  uint32_t rec_rows_a, rec_rows_z;
  int rec_ga, rec_gz;
  const uint64_t* loff0;
  const uint32_t* lidx0;
  const uint32_t* linfo;     // per list (path*M + half): padded length (multiple of 8, >= 8) | mode (bit 0: 1 = overlap list)
  const uint32_t* lover;     // per list: where entries 8.. live in `dover` (lists longer than 8)
  const uint32_t* dlist;     // per list: its first 8 entries at [list*8, list*8+8)
  const uint32_t* dover;
  const float* t32;
  const double* d64;
  const uint32_t* ladder;    // [kLadderLevels + 2][ladder_stride] hi << 16 | lo (method 1); rows kLadderLevels, +1: all inside / all outside
  uint32_t* null_bits;
  uint64_t* timing;          // diagnostics build (-DGCRE_IE_TIMING): 6 per-section cycle sums over all waves
  uint32_t* stats;           // optional: [0] += joined-path tiles that were looked up (not pruned)
  uint32_t* planes_out;      // optional: planes of the joined paths [tile][(out_first+q)*M+h][go][64][4]
  int64_t out_first;
  int64_t nsegs;
  int64_t seg_begin, seg_end;        // slice of the segment table this launch walks
  uint32_t score_begin, score_end;   // joined paths of the launch outside [begin, end) only produce planes
  const uint32_t* rec_segs;          // pruned method-1 kernel with a recipe: kRecSegWords words per segment (k_fill_rec_segs)
  int batch;                         // segments per ticket
  uint32_t* queue;                   // pruned kernels: the eight ticket counters of the launch (16 words apart), zero on entry
  uint32_t score_segs;               // the same range in segments: the table's first score_segs segments (they do not straddle)
  int nkt, waves_per_xcd, K;
  int g0, gz, go;            // plane groups (4 planes each) of the three plane arrays
  int lad_mode;              // method-1 kernel: 0 thresholds from the running maxima, 1 look nothing up, 2 look everything up
  uint32_t g00_rows;         // signed kernel: vtmax[0][0] <= g00_rows / (2 kLadderPerUnit), what an EMPTY half contributes to a
                             // path's null score (0 for the hypergeometric table); 0xffffffff: not bounded (NaN)
  int ladder_stride;         // = number of table diagonals
  uint32_t mt_rows, zoff;
  // quad form of the pruned method-1 kernel (gcre_ieq.hip): entry = first segment | (segments - 1) << 30, up to four
  // consecutive segments of `segs` that join the same paths1 rows
  const uint32_t* quads;
  int64_t quad_begin, quad_end;
};
hipError_t launch_null_ie_quad(const IeArgs& a, int planes, hipStream_t stream);   // gcre_ieq.hip
int ieq_max_waves_per_cu(int planes, int gz, bool rec);
int ieq_quad_segs();   // segments a quad may hold (gcre_ieq.hip)
// r_tot (optional): carriers of the recipe's rows; bits 1-2 of the gathered list-info word then say how many groups of 4
// count planes of the row can be non-zero, minus one (counts never exceed the carrier total)
hipError_t launch_fill_rec_segs(const SparseSeg* segs, int64_t nsegs, const uint32_t* r_row0, const uint32_t* r_rowz,
                                const uint32_t* r_linfo, const uint32_t* r_lover, const uint32_t* r_slot, const uint32_t* r_tot,
                                uint32_t* out, hipStream_t stream);
hipError_t launch_null_ie(const IeArgs& a, int method, int planes, bool general, hipStream_t stream);
int ie_max_waves_per_cu(int method, int planes, int gz, bool out, bool rec);
// the signed method's pruned kernel (gcre_ie2.hip); rec_rows_a / rec_rows_z count row-halves
hipError_t launch_null_ie_m2(const IeArgs& a, int planes, hipStream_t stream);
int ie2_max_waves_per_cu(int planes, int gz, bool out, bool rec);
int ie2_steps();   // steps of the staircase cover of F + G <= theta (GCRE_M2_STEPS)
hipError_t launch_build_planes(const uint32_t* mt, uint32_t mt_rows, int nkt, const uint64_t* loff, const uint32_t* lidx,
                               int64_t nrowhalves, int groups, uint32_t* planes, hipStream_t stream);
hipError_t launch_build_ladder(const float* t32, int TD, uint32_t* ladder, hipStream_t stream);
hipError_t launch_build_ladder2(const double* dmax, int TD, uint32_t* ladder, hipStream_t stream);   // signed method, half thresholds
// exclusive prefix sum of n u32 counts into n+1 u64 offsets; the low 2 bits of a count do not add, they are copied
// into the low bits of its offset (list lengths are multiples of 4, bit 0 carries the IE list mode) (scratch: >= (n+1023)/1024 + 1 u64)
hipError_t launch_scan_u32_u64(const uint32_t* cnt, int64_t n, uint64_t* off, uint64_t* scratch, hipStream_t stream);
// per joined path: entries of paths1's list whose bit is clear in the paths0 row, 16-padded, at dlist[doff[i]..)
hipError_t launch_delta_fill(const uint32_t* p0, int S32, int W32p, int method, const uint32_t* row0,
                             const uint32_t* row1, int64_t count, const uint64_t* loff1, const uint32_t* lidx1,
                             const uint64_t* doff, uint32_t zoff, uint32_t* dlist, hipStream_t stream);
hipError_t launch_row_bits(const uint32_t* rows, int64_t nrows, int S32, int W32p, uint32_t* cnt, hipStream_t stream);
hipError_t launch_row_fill(const uint32_t* rows, int64_t nrows, int S32, int W32p, const uint64_t* off, uint32_t zoff,
                           uint32_t* idx, hipStream_t stream);
hipError_t launch_build_mt(const uint32_t* masks, int W32p, int Kpad, int nkt, uint32_t mt_rows, uint32_t* mt,
                           hipStream_t stream);

hipError_t launch_pack_dense(const int32_t* data, int64_t nrow, int ncol, int col_major, uint64_t* rows, int S,
                             hipStream_t stream);
hipError_t launch_select(const uint64_t* from, const int32_t* idx, int64_t n, int S, uint64_t* out, hipStream_t stream);
hipError_t launch_masks_from_ints(const int32_t* perms, int nrows_in, int ncol, int col_major, const Geometry& g,
                                  uint32_t* masks, hipStream_t stream);
hipError_t launch_masks_from_words(const uint64_t* packed, int nrows_in, const Geometry& g, uint32_t* masks,
                                   hipStream_t stream);

// joined-path ordinal -> (row of paths0, row of paths1 | swap<<31)
hipError_t launch_expand(const int64_t* path_idx, const int64_t* location, int64_t n_uids, const int32_t* signs,
                         int path_length, int method, int64_t first, int64_t count, uint32_t* row0, uint32_t* row1,
                         hipStream_t stream);

struct StatsArgs {
  const uint64_t* p0;
  const uint64_t* p1;
  const uint32_t* row0;
  const uint32_t* row1;
  const uint64_t* case_mask;   // [Wp]
  const double* dvt;           // f64 value table, diagonal-major
  uint64_t* key;               // order-preserving u64 image of the real score (0 = never a candidate)
  uint32_t* tot;               // M per path
  uint32_t* cases;
  uint32_t* ctrls;
  uint64_t* res;               // kept rows, indexed by absolute ordinal (first + i), or nullptr
  uint32_t* max_tot;           // optional: running maximum of the carrier totals (sizes the sparse kernel's counters)
  uint32_t* dcnt;              // optional: popcount(path1 & ~path0) rounded up to 4, per joined path (and half)
  // inclusion-exclusion form (all optional): pz = reduced operand, zindex[row of paths1] = its row (nullptr: same row)
  const uint64_t* pz;
  const int32_t* zindex;
  uint32_t* rowz;              // out: reduced row | swap << 31 per joined path
  uint32_t* bad;               // out: set to 1 when some joined path differs from paths0 | reduced row
  int ie_bias;                 // overlap list chosen when overlap + ie_bias < delta; negative: never
  int ie_rule;                 // 1: overlap list whenever it fits the 8-entry slot or is shorter than the delta list (method 1: the
                               // quad kernel has the added row's planes at hand anyway); 0: the bias rule
  // k_stats_ie only: the lists themselves, written in the same pass (no scan, no fill kernel).  List d = path*M + half
  // has its first 8 entries in slot[d*8 .. d*8+8) and the rest, when it is longer, in over[lover[d] ..); both padded
  // with `zoff` to a multiple of 8.  linfo[d] = padded length | mode (bit 0: 1 = overlap list).
  const uint64_t* excess;      // hinted joins: [ranges][S] union of paths1 & ~reduced over each uid range (k_range_union), or nullptr
  const int32_t* range_of;     // uid (row of paths0) -> its range
  uint32_t* linfo;
  uint32_t* lover;
  uint32_t* slot;
  uint32_t* over;
  uint32_t over_cap;           // entries `over` can hold; ov_count beyond it means: grow and run again
  uint32_t* ov_count;          // entries reserved in `over` so far
  uint32_t zoff;
  // k_stats_ie2h (method 2): CSR offsets of the reduced operand's bit lists (two per row), set only when every row of it
  // has an empty half
  const uint64_t* lz_off;
  int64_t first;
  int64_t count;
  int S;
  int Wp;
};
hipError_t launch_stats(const StatsArgs& a, int method, hipStream_t stream);
hipError_t launch_stats_ie(const StatsArgs& a, int method, hipStream_t stream);   // gcre_ie.hip
hipError_t launch_range_union(const uint64_t* p1, const uint64_t* pz, const int32_t* zindex, const int32_t* pair_range,
                              const int64_t* pair_loc, int64_t npairs, int S, int Wp, int method, uint64_t* excess,
                              uint32_t* bad, hipStream_t stream);

// ---- top-k selection over key[0..count) ----
hipError_t launch_hist(const uint64_t* key, int64_t count, int shift, uint64_t prefix, uint32_t* hist256,
                       hipStream_t stream);
// all eight digit passes queued back to back, the state between them stays on the device: afterwards st->prefix is the
// need-th largest key, st->greater the number of keys in higher buckets, st->need / st->eq_count the wanted / present
// number of keys equal to it
struct SelectState {
  uint64_t prefix;
  int64_t need;
  int64_t greater;
  uint32_t eq_count;
  uint32_t pad;
};
hipError_t launch_radix_select(const uint64_t* key, int64_t count, int64_t need, uint32_t* hist256, SelectState* st,
                               hipStream_t stream);
// appends every i with key[i] > thr (any order) to out[], counter in *n_out
hipError_t launch_collect_gt(const uint64_t* key, int64_t count, uint64_t thr, uint32_t* out, uint32_t* n_out,
                             uint32_t cap, hipStream_t stream);
// per 1024-entry chunk: number of key[i] == thr
hipError_t launch_eq_count(const uint64_t* key, int64_t count, uint64_t thr, uint32_t* chunk_cnt, hipStream_t stream);
// first `m` (in index order) entries with key[i] == thr, written at out[rank]; chunk_base = exclusive scan of chunk_cnt
hipError_t launch_eq_collect(const uint64_t* key, int64_t count, uint64_t thr, const uint32_t* chunk_base, uint32_t m,
                             uint32_t* out, hipStream_t stream);
hipError_t launch_gather_winners(const uint32_t* sel, uint32_t nsel, const uint64_t* key, const uint32_t* cases,
                                 const uint32_t* ctrls, const uint32_t* row0, const uint32_t* row1, uint64_t* o_key,
                                 uint32_t* o_cases, uint32_t* o_ctrls, uint32_t* o_row0, uint32_t* o_row1,
                                 hipStream_t stream);
hipError_t launch_generate_masks(uint64_t seed, int K, int n, int n_strata, const int32_t* stratum, const uint32_t* cases_in,
                                 const uint32_t* size_of, uint32_t* work, int W32p, int Kpad, uint32_t* masks,
                                 hipStream_t stream);
hipError_t launch_table_to_diag(const double* table, int nrow, int ncol, int col_major, int n, int TD, double* dvt,
                                float* t32, double* dmax, hipStream_t stream);
hipError_t launch_fill_u32(uint32_t* p, int64_t n, uint32_t v, hipStream_t stream);
hipError_t launch_max_merge(uint32_t* dst, const uint32_t* src, int n, hipStream_t stream);

}  // namespace gcre
