/*
 * gcre_hip.h -- C ABI of libgcre_hip.so, the MI355X (gfx950) implementation of geneticsCRE's
 * permutation-tested path-join scorer.
 *
 * This is the drop-in boundary: the entry points below are what a replacement for the reference's
 * src/wrapper.cpp binds (R .Call shim) and what the Python host in
 * geneticscre_amd/api.py binds through ctypes.  Plain pointers and sizes only; nothing throws across
 * the boundary; every function returns 0 on success or a negative gcre_status, and the message is
 * available from gcre_last_error().  A context is used from one host thread at a time.
 *
 * Reference interface replaced (paths relative to /root/reference):
 *   JoinExec ctor / setValueTable / setPermutedCases   src/join_base.cpp:37-125, src/gcre.h:103-180
 *   PathSet ctor / load / select                        src/gcre_paths.h:19-92
 *   JoinExec::join (+ JoinMethod1/2::score_permute)     src/join_base.cpp:189-264, src/methods.h:58-232
 *   ProcessPaths (the 39-argument driver)               src/wrapper.cpp:177-281
 */
#ifndef GCRE_HIP_H
#define GCRE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCRE_ABI_VERSION 4

typedef enum {
  GCRE_OK = 0,
  GCRE_ERR_ASSERT = -1,   /* std::logic_error("assertion") in the reference (gcre_types.h:58-66) */
  GCRE_ERR_RANGE = -2,    /* std::out_of_range("assertion")            (gcre_types.h:68-76) */
  GCRE_ERR_DEVICE = -3,   /* HIP runtime failure or no gfx950 device */
  GCRE_ERR_ARG = -4       /* NULL / malformed argument */
} gcre_status;

typedef struct gcre_ctx gcre_ctx;           /* JoinExec   -- src/gcre.h:103-180 */
typedef struct gcre_pathset gcre_pathset;   /* PathSet    -- src/gcre_paths.h:10-98, rows live in HBM */

/* joined_res + Score -- src/gcre_types.h:32-48.  Arrays are owned by the library until gcre_result_free. */
typedef struct {
  int32_t n;            /* entries, <= top_k, ascending score; may hold the {-inf,-1,-1,0,0} sentinel
                           when the level has fewer than top_k scorable paths (join_base.cpp:194) */
  double* scores;
  int32_t* src;         /* Score.src = row of paths0 / uid index (methods.h:92) */
  int32_t* trg;         /* Score.trg = row of paths1 */
  int32_t* cases;
  int32_t* ctrls;
  int32_t n_perm;       /* = iterations requested */
  float* null_max;      /* per-permutation maximum over this level's paths, f32 (methods.h:101-102) */
} gcre_result;

/* Optional knobs of one join.  Zero-initialise for the reference behaviour. */
typedef struct {
  int32_t sharded;      /* 0: score every joined path (reference behaviour); 1: only [shard_begin, shard_end) */
  int32_t keep_ranged;  /* 0: kept path rows are produced in full, whatever the shard; 1: only rows [keep_begin, keep_end)
                           and the scored shard are produced -- the rest of `res` is left untouched; 2: every row is
                           produced, but only rows [keep_begin, keep_end) and the shard get their permutation count planes
                           (what a later join needs of the rows it reads as paths0; any other use rebuilds them).  For
                           multi-device runs: a device only needs the kept rows its own shards of the later joins read */
  int64_t shard_begin;  /* joined-path ordinal range scored on THIS device (may be empty). */
  int64_t shard_end;
  void* d_null_out;     /* optional device pointer to iterations floats: receives this shard's null maxima
                           (for an RCCL MAX all-reduce by the caller); may be NULL */
  int64_t keep_begin;   /* joined-path ordinals = rows of `res` (see keep_ranged) */
  int64_t keep_end;
  /* Multi-device runs: a shard prunes its table look-ups against the running per-permutation maxima (the reference's
   * `perm_scores[r] = max(...)`, src/methods.h:101-102, is what they bound), and a device that sees 1/N of the paths
   * knows lower maxima than the whole level has.  When `exchange` is set the join calls it exactly `exchanges` times --
   * after the warm-up slice and between the slices (1/2^(E-1), .., 1/4, 1/2 of the shard) of its permutation kernel --
   * with the device buffer d_null_out holding this shard's maxima of permutations [k0, k1) so far (floats, >= 0); the
   * callee MAX-all-reduces them in place across the devices (RCCL) and returns 0 once the buffer may be read; the join
   * goes on with the merged maxima as thresholds.  Every device must pass the same `exchanges`; a join that takes another
   * kernel form still makes its calls, so that the collectives match up.  Results do not depend on it.  d_null_out is
   * required. */
  int32_t exchanges;
  int (*exchange)(void* user, void* d_null, int32_t k0, int32_t k1);
  void* exchange_user;
} gcre_join_opts;

/* Timing of the last join / process_paths call, measured with HIP events on the library's stream. */
typedef struct {
  double null_kernel_ms;      /* sum over launches of the permutation (null) kernel */
  int64_t null_kernel_launches;
  double stats_kernel_ms;     /* real-label scoring + kept-row materialisation */
  double select_ms;           /* top-k selection */
  double total_ms;            /* whole device region */
  int64_t paths;              /* joined paths scored on this device */
  int64_t scores;             /* paths x iterations */
  double null_alg_bytes;      /* algorithmic HBM bytes of the null kernel launches */
  double null_row_loads;      /* sparse kernel: mask-row loads issued (list entries x permutation tiles); 0 = dense kernel */
  int64_t ie_launches;        /* launches of the inclusion-exclusion kernel (gcre_ie.hip) */
  int64_t ie_overlap_lists;   /* joined-path halves scored as N0 + Nz - overlap (the rest streamed their delta list) */
  int64_t ie_hinted_joins;    /* joins that ran on a verified gcre_uids_set_reduced operand */
  int64_t ie_plane_joins;     /* joins whose paths0 count planes were already resident */
  int64_t ie_lookup_tiles;    /* joined-path x 2048-permutation tiles that survived the pruning test (method 1) */
  double prepare_ms;          /* host wall time: bit lists / count planes of the operands (once per set and mask epoch) */
  double inspect_ms;          /* host wall time: list offsets (scan), list fill and the syncs around them */
  int64_t ie_quad_launches;   /* of ie_launches: pruned method-1 launches that ran the four-paths-per-wave form (gcre_ieq.hip) */
  int64_t inspect_replays;    /* chunks that started at the null kernel: their inspector output was still in place (gcre_set_inspect_cache) */
} gcre_profile;

/* ---- context: JoinExec::JoinExec, src/join_base.cpp:37-59.  method 1 = unsigned, 2 = signed. ---- */
gcre_ctx* gcre_create(int method, int n_cases, int n_ctrls, int iterations, int device);
void gcre_destroy(gcre_ctx* ctx);   /* also releases every path set and uids object still alive on the context: their handles die with it */
const char* gcre_last_error(const gcre_ctx* ctx);   /* ctx may be NULL: error of the last failed gcre_create */
int gcre_abi_version(void);
/* the extra compiler flags the library was built with ("" for the shipped build): a diagnostics build that switches parts of
   a kernel off for a timing experiment (GCRE_*_NO*, results are wrong) is recognisable, __graft_entry__.smoke() asserts "" */
const char* gcre_build_flags(void);
int gcre_device_count(void);   /* gfx950 devices visible to the process (0 when there is none: gcre_create then fails) */

int gcre_set_top_k(gcre_ctx* ctx, int top_k);        /* JoinExec::top_k, src/gcre.h:120 (default 12) */
int gcre_width_ul(const gcre_ctx* ctx);              /* 64-bit words per case/control mask, ceil(n/64) */
int gcre_vlen(const gcre_ctx* ctx);                  /* words per path row as seen by the host = width * method */

/* JoinExec::setValueTable, src/join_base.cpp:62-80.  nrow x ncol doubles; col_major != 0 for an R matrix. */
int gcre_set_value_table(gcre_ctx* ctx, const double* table, int nrow, int ncol, int col_major);

/* JoinExec::setPermutedCases, src/join_base.cpp:85-125.  nrow x ncol ints, 1 = label kept. */
int gcre_set_perm_cases(gcre_ctx* ctx, const int32_t* perms, int nrow, int ncol, int col_major);

/* Extension: the same masks already packed, nrow x width_ul words, bit c of row r = "patient c is a case
 * under permutation r" (what setPermutedCases derives).  Same row reuse / truncation rules. */
int gcre_set_perm_masks(gcre_ctx* ctx, const uint64_t* masks, int nrow);

/* ---- path sets ---- */
gcre_pathset* gcre_pathset_zeros(gcre_ctx* ctx, int64_t nrows);                    /* createPathSet, join_base.cpp:157-161 */
gcre_pathset* gcre_pathset_from_dense(gcre_ctx* ctx, const int32_t* data, int64_t nrow, int ncol,
                                      int col_major);                               /* PathSet::load, gcre_paths.h:56-78 */
gcre_pathset* gcre_pathset_from_words(gcre_ctx* ctx, const uint64_t* rows, int64_t nrows);   /* packed rows, vlen words each */
gcre_pathset* gcre_pathset_select(gcre_ctx* ctx, const gcre_pathset* from, const int32_t* idx,
                                  int64_t n);                                       /* PathSet::select, gcre_paths.h:82-92 */
int64_t gcre_pathset_size(const gcre_pathset* ps);
int gcre_pathset_read(gcre_ctx* ctx, const gcre_pathset* ps, uint64_t* out_rows);   /* device -> host, size x vlen words */
void gcre_pathset_free(gcre_pathset* ps);

/*
 * JoinExec::join, src/join_base.cpp:189-264.
 *   uid_count / uid_location : uid_ref.count / .location per row of paths0 (src/gcre_types.h:50-56)
 *   signs                    : UidRelSet::signs, used by method 2 through need_flip (src/gcre.h:71-81)
 *   res                      : receives the joined rows (size must equal the total path count) or NULL
 * Ties between equal scores are resolved towards the smaller joined-path ordinal ("Ties").
 */
int gcre_join(gcre_ctx* ctx, int path_length,
              const int32_t* uid_count, const int64_t* uid_location, int64_t n_uids,
              const int32_t* signs, int64_t n_signs,
              const gcre_pathset* paths0, const gcre_pathset* paths1, gcre_pathset* res,
              const gcre_join_opts* opts, gcre_result* out);
void gcre_result_free(gcre_result* r);

/* UidRelSet (src/gcre.h:49-90) kept resident on the device, for callers that join the same level repeatedly
 * (benchmarks, sharded runs): gcre_join_uids == gcre_join without re-uploading the index. */
typedef struct gcre_uids gcre_uids;
gcre_uids* gcre_uids_create(gcre_ctx* ctx, int path_length, const int32_t* uid_count, const int64_t* uid_location,
                            int64_t n_uids, const int32_t* signs, int64_t n_signs);
int64_t gcre_uids_total_paths(const gcre_uids* uids);   /* UidRelSet::count_total_paths, gcre.h:83-88 */
void gcre_uids_free(gcre_uids* uids);
/* Optional speed hint, no reference counterpart (the reference walks every word of both operands, methods.h:73-88).
 * States that for every joined path (idx, loc) of this index
 *     paths0[idx] | paths1[loc]  ==  paths0[idx] | reduced[index[loc]]
 * e.g. at path length 4 paths1[loc] is the 2-gene path (c, d) whose first gene c already lies on paths0[idx], so
 * `reduced` = the per-gene rows and index[loc] = d.  The null kernel then adds the reduced row's precomputed
 * permutation counts instead of walking paths1[loc].  The claim is checked on the device for every join
 * (reduced row inside the joined row, equal carrier totals); a join for which it fails silently runs on paths1.
 * `index` has one entry per row of paths1 (n >= largest location + 1), values in [0, rows(reduced)); bit 31 of an
 * entry set = (signed method) the reduced row enters with its (+)/(-) halves swapped relative to how paths1[loc]
 * enters (UidRelSet::need_flip, gcre.h:71-81).
 * The caller keeps `reduced` alive while the index is used.  reduced == NULL removes the hint. */
int gcre_uids_set_reduced(gcre_uids* uids, const gcre_pathset* reduced, const int32_t* index, int64_t n);
int gcre_join_uids(gcre_ctx* ctx, const gcre_uids* uids, const gcre_pathset* paths0, const gcre_pathset* paths1,
                   gcre_pathset* res, const gcre_join_opts* opts, gcre_result* out);

/* Inspect-ahead.  A join is an inspector (expansion, real-label statistics and observed scores, kept rows, the lists the
 * permutation kernel streams, top-k selection: nothing of it depends on the permutation masks) followed by its permutation
 * kernel, and the inspector of the NEXT join of a sequence (src/wrapper.cpp:225-276) only reads what THIS join's inspector
 * wrote -- the kept rows -- not what its permutation kernel computes.  gcre_join_ahead registers the next join: the following
 * gcre_join / gcre_join_uids call on the context runs that join's inspector on a stream of its own as soon as its own kernels
 * are in flight, into the registered join index's inspection cache, and the registered join -- the same index, operands, kept
 * set and opts, called next -- starts at its permutation kernel.  Needs the inspection cache (gcre_set_inspect_cache(ctx, 1)
 * for the pass) and GCRE_AHEAD=1 in the environment: it is OFF by default -- measured, it does not pay (both kernels fill the
 * GPU) -- and without either the call registers nothing and every join runs whole.  uids = NULL
 * cancels a registration; freeing a registered object cancels it too.  Results never depend on it. */
int gcre_join_ahead(gcre_ctx* ctx, const gcre_uids* uids, const gcre_pathset* paths0, const gcre_pathset* paths1,
                    gcre_pathset* res, const gcre_join_opts* opts);


int gcre_get_profile(const gcre_ctx* ctx, gcre_profile* out);

/*
 * ProcessPaths, src/wrapper.cpp:177-281: the whole six-join sequence on plain arrays.
 * Level order of the six uid tables: 1a, 1b, 2, 3, 4, 5 (wrapper.cpp:177-182).  uid_count/uid_location are the
 * already-resolved count_locs (wrapper.cpp:106-132; gcre_resolve_count_locs below does that lookup).
 */
typedef struct {
  const int32_t* uid_count;
  const int64_t* uid_location;
  int64_t n_uids;
  const int32_t* signs;
  int64_t n_signs;
} gcre_level;

typedef struct {
  gcre_level level[6];
  const int32_t* data_inds[4];   /* 1a, 1b, 2, 3 -- 0-based rows of data1 / data2 (wrapper.cpp:205-208) */
  int64_t n_data_inds[4];
  const int32_t* data1;          /* genes x patients */
  int64_t data1_rows;
  const int32_t* data2;
  int64_t data2_rows;
  int data_col_major;            /* R matrices are column-major */
  const double* value_table;
  int vt_rows, vt_cols, vt_col_major;
  const int32_t* perm_cases;     /* iterations x patients; NULL with 0 rows = keep the masks the context already
                                  * holds (gcre_generate_perm_masks / gcre_set_perm_masks), an error if it has none */
  int perm_rows, perm_col_major;
  int path_length;               /* 1..5 */
  /* one device of several (gcre_process_paths_devices fills these; 0 / 0 / 0 = the whole job on this context): the
   * context scores joined-path ordinals [P * shard_rank / shard_world, P * (shard_rank + 1) / shard_world) of every
   * level and keeps every row; window_perms > 0 fixes the permutation window (all devices must walk the same ones) */
  int shard_rank, shard_world, window_perms;
} gcre_pp_input;

/* out[0..4] = lst1..lst5; entries above path_length have n = -1 (R sees NULL, wrapper.cpp:223). */
int gcre_process_paths(gcre_ctx* ctx, const gcre_pp_input* in, gcre_result out[5]);

/* The same call on several GPUs of one node from ONE process -- what the .Call shim uses, so that the drop-in takes the
 * node like the reference takes `nthreads` cores (src/join_base.cpp:163-185, src/wrapper.cpp:189).  One context and one
 * host thread per device; every device scores 1/n of each level's joined paths (contiguous ordinals, src/join_base.cpp:230
 * workers pull uids the same way) and keeps every row; the per-permutation null maxima are MAX-merged and the top-k
 * tables merged with the sentinel rule on the host when the devices are done (K floats + top_k rows per level and
 * device: the message sizes of SURVEY.md 2.2).  devices = NULL: devices 0 .. n_devices-1; n_devices <= 0: every visible
 * device; a device may be listed twice (rehearsal on one GPU).  Results are bit-identical for any device list.
 * err / errlen: optional buffer for the failing device's message. */
int gcre_process_paths_devices(int method, int n_cases, int n_ctrls, int iterations, int top_k, const int* devices,
                               int n_devices, const gcre_pp_input* in, gcre_result out[5], char* err, size_t errlen);

/* RCCL inside gcre_process_paths_devices.  When every device is listed once and librccl.so loads (dlopen: the library does
 * not link it), each device gets a communicator (ncclCommInitAll) and the K-float null maxima are merged by
 * ncclAllReduce(ncclFloat32, ncclMax) in place on the devices -- once per level and permutation window (merge_scores,
 * src/methods.h:34-37) and at every threshold exchange inside a large join; the top-k tables (top_k rows per device) are
 * merged on the host.  Environment: GCRE_RCCL=0 host-side merge only, GCRE_RCCL=force communicators for a single device
 * too (one-rank collectives).  gcre_rccl_selftest: a one-rank communicator on `device`, one MAX all-reduce, checked;
 * gcre_rccl_collectives: RCCL collectives this process has issued so far. */
int gcre_rccl_selftest(int device, char* err, size_t errlen);
int64_t gcre_rccl_collectives(void);

/* uid resolution of assemble_uids (src/wrapper.cpp:106-132): row k takes (count, location) of the entry
 * keyed by trg_uids[k]; missing keys give (0, 0). */
int gcre_resolve_count_locs(const int32_t* trg_uids, int64_t n_uids,
                            const int32_t* keys, const int32_t* counts, const int32_t* locations, int64_t n_keys,
                            int32_t* out_count, int64_t* out_location);

/* ------------------------------------------------------------------------------------------------------------
 * Callers and data formats either side of the path: native versions of what GWASPA prepares in R.
 * Not needed by the .Call drop-in (R keeps doing this work); used by non-R front ends and by the benchmark.
 * ------------------------------------------------------------------------------------------------------------ */

/* One join level as GWASPA builds it (R/ProcessPaths.R:214-256): uid_ref rows + UidRelSet::signs. */
typedef struct {
  int32_t path_length;
  int64_t n_uids;
  int32_t* src;         /* uid_ref.src */
  int32_t* trg;         /* uid_ref.trg */
  int32_t* count;       /* uid_ref.count */
  int64_t* location;    /* uid_ref.location (-1 where count == 0, R/PathMethods.R:147) */
  int64_t n_signs;
  int32_t* signs;
  int64_t total_paths;  /* UidRelSet::count_total_paths */
} gcre_level_table;

typedef struct {
  gcre_level_table level[6];   /* 1a, 1b, 2, 3, 4, 5 */
  int64_t n_data_inds[4];
  int32_t* data_inds[4];       /* 1a, 1b, 2, 3: 0-based rows of data1 / data2 */
  int64_t n_rels3;             /* getRels3 (src/wrapper.cpp:18-48): one row per 2-edge walk */
  int32_t *r3_src, *r3_trg, *r3_sign, *r3_trg2, *r3_sign2;
} gcre_levels;

/* Relations must be sorted by (src, trg), unique, without self loops; gene ids are 0..n_genes-1 = rows of the data
 * matrix (what GWASPA's filtering leaves, R/ProcessPaths.R:133-167, 210).  Arrays are malloc'ed; free with
 * gcre_levels_free. */
int gcre_build_levels(int32_t n_genes, const int32_t* src, const int32_t* trg, const int32_t* sign, int64_t n_edges,
                      gcre_levels* out);
void gcre_levels_free(gcre_levels* levels);

/* getValuesTable (R/Utils.R:137-159): out[(n_cases+1) x (n_ctrls+1)] row-major, -log two-sided hypergeometric p.
 * Parity with R's stats::dhyper is unpinned (no R in the build image); the scorer treats the table as opaque input. */
int gcre_values_table(int n_cases, int n_ctrls, double* out);
/* Which summation order gcre_values_table uses at this size: 1 = R's index order for every cell (the table R builds wherever
 * libm agrees), 0 = the sorted prefix sum of very large cohorts (past 2.5e11 inner steps, ~14,000 patients: same outcomes, same
 * accumulator, the rounded double can move in its last place).  Fixtures record it next to their input digest.  The test-only
 * GCRE_VT_EXACT_WORK environment variable moves the threshold for both functions alike. */
int gcre_values_table_exact_order(int n_cases, int n_ctrls);

/* getRandIndicesMat + getCaseORControl + setPermutedCases (R/Utils.R:22-46, 246-262; src/join_base.cpp:85-125) fused
 * on the device: permutation r = a uniformly random relabelling that keeps n_cases cases (inside every stratum when
 * `stratum[n]` is given, values 0..n_strata-1, R/Utils.R:8-13).  Deterministic in (seed, r, patient): see
 * gcre_mix64 and k_generate_masks.  Replaces gcre_set_perm_cases for callers that do not need R's RNG stream. */
int gcre_generate_perm_masks(gcre_ctx* ctx, uint64_t seed, const int32_t* stratum, int n_strata);
uint64_t gcre_mix64(uint64_t z);
/* Restrict the joins that follow to permutations [k0, k1) (k0 a multiple of 2048; k1 a multiple of 2048 or = iterations).
 * A join then returns k1 - k0 null maxima; observed scores and top-k lists do not depend on the window.  Lets a caller
 * run a large permutation count in batches whose count planes (one per kept row and 2048-permutation tile) fit in device
 * memory -- gcre_process_paths does so on its own.  Setting masks resets the window to [0, iterations). */
int gcre_set_perm_window(gcre_ctx* ctx, int k0, int k1);
/* Window length (permutations; a multiple of 2048, or iterations when everything fits) for a pipeline whose path sets --
 * the inputs and the kept sets -- hold set_rows[i] rows: the count planes of a set take up to 4 KB per row, method half
 * and tile (method 1: sets above the recipe limit, GCRE_PLANES_OUT_MAX_MB, store none) and should leave half of the free
 * device memory alone. */
int gcre_plan_perm_window(gcre_ctx* ctx, const int64_t* set_rows, int n_sets);
/* Inspection cache.  What a join computes before its permutation (null) kernel -- expanded row numbers, carrier
 * totals, observed scores and their top-k, the kept rows, the inclusion-exclusion lists (JoinExec::join's real-label half,
 * src/join_base.cpp:236-262 + methods.h:90-99) -- does not depend on the permutation masks.  With the cache on, that
 * output stays with the join index (gcre_uids) it was computed for, and a later join on the same index with the same
 * operand rows, kept set, shard, top_k and value table starts at the null kernel: the 2nd..nth permutation window of a
 * large run, or the next pass over resident inputs.  Results are identical either way.  gcre_process_paths turns it on
 * by itself for a call that needs more than one window.  Off by default (the buffers cost ~80 B per joined path).
 * gcre_drop_inspections forgets what is cached (release_memory = 0 keeps the buffers for the next run). */
int gcre_set_inspect_cache(gcre_ctx* ctx, int on);
int gcre_drop_inspections(gcre_ctx* ctx, int release_memory);
/* read permutation mask r back as width_ul words (bit c = patient c is a case under permutation r) */
int gcre_get_perm_mask(gcre_ctx* ctx, int r, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif
