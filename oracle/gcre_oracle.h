/*
 * gcre_oracle.h -- CPU restatement of geneticsCRE's permutation-tested path-join scorer.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker ("oracle") for the HIP path in
 * geneticscre_amd/.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * may load it; the product never routes through it.
 *
 * Parity status: PINNED by the known-answer vectors of SURVEY.md Appendix B (captured from the
 * unmodified reference binary) -- tests/golden/appendix_b.json -- and nothing else: the reference
 * ships no golden vectors and its join engine (src/join_base.cpp:4) includes <Rcpp.h>, which this
 * image lacks, so the reference itself is not buildable here ("Oracle").
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 */
#ifndef GCRE_ORACLE_H
#define GCRE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gcre_o_ctx gcre_o_ctx;

/* JoinExec::JoinExec -- src/join_base.cpp:37-59.  method: 1 | 2.  Returns NULL on bad args. */
gcre_o_ctx* gcre_o_create(int method, int n_cases, int n_ctrls, int iters);
void gcre_o_destroy(gcre_o_ctx* ctx);

/* words per case/control mask (our own padding: ceil(n/64); layout only) */
int gcre_o_width(const gcre_o_ctx* ctx);
/* words per stored path row = width * method -- src/join_base.cpp:157-161 */
int gcre_o_vlen(const gcre_o_ctx* ctx);

/* JoinExec::setValueTable -- src/join_base.cpp:62-80.  tbl row-major nrow x ncol. */
int gcre_o_set_value_table(gcre_o_ctx* ctx, const double* tbl, int nrow, int ncol);

/* JoinExec::setPermutedCases -- src/join_base.cpp:85-125.  perms row-major nrow x ncol, 1 = label kept. */
int gcre_o_set_perm_cases(gcre_o_ctx* ctx, const int* perms, int nrow, int ncol);

/* the same masks supplied already packed: nrow x width words, bit c = patient c is a case under permutation r */
int gcre_o_set_perm_masks(gcre_o_ctx* ctx, const uint64_t* masks, int nrow);

/* read back permutation mask r as width words (test hook) */
int gcre_o_get_perm_mask(const gcre_o_ctx* ctx, int r, uint64_t* out);

/* PathSet::load -- src/gcre_paths.h:56-78.  data row-major nrow x ncol; out is nrow x vlen, zeroed here. */
int gcre_o_pack_dense(const gcre_o_ctx* ctx, const int* data, int nrow, int ncol, uint64_t* out);

/*
 * JoinExec::join -- src/join_base.cpp:189-264 with JoinMethod1/2::score_permute
 * (src/methods.h:58-105, 130-232), merge_scores (methods.h:25-39), format_result (join_base.cpp:138-154).
 *
 *   uid_count/uid_location : per uid row (uid_ref.count / .location, src/gcre_types.h:50-56)
 *   signs                  : UidRelSet::signs, need_flip rule src/gcre.h:71-81
 *   paths0 [n0 x vlen], paths1 [n1 x vlen]; paths_res [total_paths x vlen] or NULL (no keep)
 *   order_mode 0 : emulate the reference's std::priority_queue arrival order (nthreads as given)
 *   order_mode 1 : canonical order (score desc, then path index asc) -- SURVEY App. A-9
 * Outputs (caller-allocated):
 *   out_scores/src/trg/cases/ctrls : capacity top_k; ascending score; *out_n entries filled
 *                                    (may include the {-inf,-1,-1,0,0} sentinel, App. A-8)
 *   out_null                       : iters floats (f32 null maxima, App. A-7)
 *   all_scores/all_cases/all_ctrls : optional (may be NULL), one entry per joined path in path order
 * Returns 0, or a negative error code (-1 assertion as in check_true/check_equal, -2 out_of_range).
 */
int gcre_o_join(gcre_o_ctx* ctx, int path_length,
                const int* uid_count, const int64_t* uid_location, int64_t n_uids,
                const int* signs, int64_t n_signs,
                const uint64_t* paths0, int64_t n0,
                const uint64_t* paths1, int64_t n1,
                uint64_t* paths_res,
                int top_k, int nthreads, int order_mode,
                double* out_scores, int* out_src, int* out_trg, int* out_cases, int* out_ctrls, int* out_n,
                float* out_null,
                double* all_scores, int* all_cases, int* all_ctrls);

#ifdef __cplusplus
}
#endif
#endif
