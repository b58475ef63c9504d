/*
 * r_shim.c -- replacement for the reference's src/wrapper.cpp + src/RcppExports.cpp: a thin .Call shim
 * that unmarshals the SEXPs, dlopen()s libgcre_hip.so and forwards to the C ABI of include/gcre_hip.h.
 *
 * Built INSIDE the R package in place of the Rcpp sources:
 *     R CMD SHLIB -o geneticsCRE.so r_shim.c -ldl
 * It needs only R's own C API (no Rcpp).  R is not installed in the build container of this repository: there the
 * file is only syntax- and type-checked against declarations of the R API entry points it uses
 * (tests/r_api_decls/, tests/test_host_logic.py); it is deliberately plain C with no logic beyond marshalling.
 *
 * The unmodified R package passes doubles for several "integer" arguments (rep(1, n) sign vectors, match(...) - 1
 * index vectors, R/ProcessPaths.R:214-256); Rcpp's IntegerVector / IntegerMatrix coerce silently, so every argument
 * is coerced here (Rf_coerceVector keeps the dim attribute) before its data pointer is taken.
 *
 * Exported registration (identical to reference src/RcppExports.cpp:85-95):
 *     _geneticsCRE_getRels3 (4 args), _geneticsCRE_getMatchingList (3), _geneticsCRE_ProcessPaths (39)
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Rdynload.h>

#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gcre_hip.h"

/* ---- the C ABI, resolved at first use ---------------------------------------------------------- */
static struct {
  void* handle;
  int (*process_paths_devices)(int, int, int, int, int, const int*, int, const gcre_pp_input*, gcre_result[5], char*, size_t);
  void (*result_free)(gcre_result*);
  int (*device_count)(void);
  int (*resolve)(const int32_t*, int64_t, const int32_t*, const int32_t*, const int32_t*, int64_t, int32_t*, int64_t*);
} G;

static void load_abi(void) {
  if (G.handle) return;
  const char* path = getenv("GCRE_HIP_LIB");           /* default: next to the package's shared object */
  if (!path) path = "libgcre_hip.so";
  G.handle = dlopen(path, RTLD_NOW | RTLD_LOCAL);
  if (!G.handle) Rf_error("geneticsCRE: cannot load %s: %s", path, dlerror());
#define SYM(field, name)                                             \
  do {                                                               \
    *(void**)(&G.field) = dlsym(G.handle, name);                     \
    if (!G.field) Rf_error("geneticsCRE: %s lacks %s", path, name);  \
  } while (0)
  SYM(process_paths_devices, "gcre_process_paths_devices");
  SYM(result_free, "gcre_result_free");
  SYM(device_count, "gcre_device_count");
  SYM(resolve, "gcre_resolve_count_locs");
#undef SYM
}

/* arguments as the types the C ABI wants, whatever numeric type R handed over; protected until UNPROTECT(*np) */
static SEXP as_int(SEXP x, int* np) {
  if (TYPEOF(x) == INTSXP) return x;
  SEXP y = PROTECT(Rf_coerceVector(x, INTSXP));
  (*np)++;
  return y;
}
static SEXP as_real(SEXP x, int* np) {
  if (TYPEOF(x) == REALSXP) return x;
  SEXP y = PROTECT(Rf_coerceVector(x, REALSXP));
  (*np)++;
  return y;
}

/* named list "uid" -> c(count, location)  ==>  parallel key/count/location arrays (wrapper.cpp:106-112) */
static void flatten_count_locs(SEXP lst, int32_t** keys, int32_t** counts, int32_t** locs, R_xlen_t* n) {
  SEXP names = Rf_getAttrib(lst, R_NamesSymbol);
  *n = XLENGTH(lst);
  *keys = (int32_t*)R_alloc(*n ? *n : 1, sizeof(int32_t));
  *counts = (int32_t*)R_alloc(*n ? *n : 1, sizeof(int32_t));
  *locs = (int32_t*)R_alloc(*n ? *n : 1, sizeof(int32_t));
  for (R_xlen_t i = 0; i < *n; i++) {
    SEXP cl = VECTOR_ELT(lst, i);
    (*keys)[i] = atoi(CHAR(STRING_ELT(names, i)));      /* stoi(uid), wrapper.cpp:111 */
    if (XLENGTH(cl) < 2) Rf_error("geneticsCRE: count_locs entries must be c(count, location)");
    if (TYPEOF(cl) == INTSXP) {
      (*counts)[i] = INTEGER(cl)[0];
      (*locs)[i] = INTEGER(cl)[1];
    } else {                                            /* c(0, -1) built from doubles (R/PathMethods.R:146-148) */
      (*counts)[i] = (int32_t)Rf_asReal(cl);
      (*locs)[i] = (int32_t)REAL(Rf_coerceVector(cl, REALSXP))[1];
    }
  }
}

/* assemble_uids (wrapper.cpp:99-140) for one level */
static void fill_level(gcre_level* lv, SEXP trg_uids, SEXP count_locs, SEXP signs, int* np) {
  int32_t *keys, *counts, *locs;
  trg_uids = as_int(trg_uids, np);
  signs = as_int(signs, np);
  R_xlen_t nk, n = XLENGTH(trg_uids);
  flatten_count_locs(count_locs, &keys, &counts, &locs, &nk);
  int32_t* oc = (int32_t*)R_alloc(n ? n : 1, sizeof(int32_t));
  int64_t* ol = (int64_t*)R_alloc(n ? n : 1, sizeof(int64_t));
  if (G.resolve(INTEGER(trg_uids), n, keys, counts, locs, nk, oc, ol) != GCRE_OK)
    Rf_error("geneticsCRE: gcre_resolve_count_locs failed");
  lv->uid_count = oc;
  lv->uid_location = ol;
  lv->n_uids = n;
  lv->signs = INTEGER(signs);
  lv->n_signs = XLENGTH(signs);
}

/* make_score_list (wrapper.cpp:142-174) */
static SEXP score_list(const gcre_result* r) {
  const int m = r->n, K = r->n_perm;
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 6));
  SEXP nms = PROTECT(Rf_allocVector(STRSXP, 6));
  static const char* names[6] = {"scores", "ids", "TestScores", "cases", "controls", "debug"};
  for (int i = 0; i < 6; i++) SET_STRING_ELT(nms, i, Rf_mkChar(names[i]));
  SEXP scores = PROTECT(Rf_allocVector(REALSXP, m));
  SEXP ids = PROTECT(Rf_allocMatrix(INTSXP, m, 2));
  SEXP test = PROTECT(Rf_allocVector(REALSXP, K));
  SEXP cases = PROTECT(Rf_allocVector(REALSXP, m));
  SEXP ctrls = PROTECT(Rf_allocVector(REALSXP, m));
  SEXP debug = PROTECT(Rf_allocVector(STRSXP, m));
  for (int k = 0; k < K; k++) REAL(test)[k] = (double)r->null_max[k];   /* f32 maxima widened, wrapper.cpp:146-147 */
  for (int k = 0; k < m; k++) {
    char buf[96];
    REAL(scores)[k] = r->scores[k];
    INTEGER(ids)[k] = r->src[k] + 1;                    /* 1-based for R, wrapper.cpp:157-159 */
    INTEGER(ids)[k + m] = r->trg[k] + 1;
    REAL(cases)[k] = r->cases[k];
    REAL(ctrls)[k] = r->ctrls[k];
    snprintf(buf, sizeof buf, "[debug] %d:%d %d/%d", r->src[k], r->trg[k], r->cases[k], r->ctrls[k]);   /* :163 */
    SET_STRING_ELT(debug, k, Rf_mkChar(buf));
  }
  SET_VECTOR_ELT(out, 0, scores);
  SET_VECTOR_ELT(out, 1, ids);
  SET_VECTOR_ELT(out, 2, test);
  SET_VECTOR_ELT(out, 3, cases);
  SET_VECTOR_ELT(out, 4, ctrls);
  SET_VECTOR_ELT(out, 5, debug);
  Rf_setAttrib(out, R_NamesSymbol, nms);
  UNPROTECT(8);
  return out;
}

/* The five native results turned into the R list, and the ONE place that frees them.  Every R allocation below can leave
 * through a longjmp (allocation failure, interrupt): R_ExecWithCleanup runs release_results on that road too, so no
 * gcre_result outlives the call whatever happens while the list is built. */
static SEXP build_result_list(void* p) {
  gcre_result* res = (gcre_result*)p;
  /* list(lst1 = ..., ..., lst5 = ...); levels above path_length stay NULL (wrapper.cpp:223) */
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 5));
  SEXP nms = PROTECT(Rf_allocVector(STRSXP, 5));
  static const char* names[5] = {"lst1", "lst2", "lst3", "lst4", "lst5"};
  for (int i = 0; i < 5; i++) {
    SET_STRING_ELT(nms, i, Rf_mkChar(names[i]));
    if (res[i].n >= 0) SET_VECTOR_ELT(out, i, score_list(&res[i]));
  }
  Rf_setAttrib(out, R_NamesSymbol, nms);
  UNPROTECT(2);
  return out;
}

static void release_results(void* p) {
  gcre_result* res = (gcre_result*)p;
  for (int i = 0; i < 5; i++)
    if (res[i].n >= 0) {
      G.result_free(&res[i]);
      res[i].n = -1;
    }
}

/* ProcessPaths -- same 39 arguments, same order as src/wrapper.cpp:177-185 / R/RcppExports.R:12-14 */
SEXP _geneticsCRE_ProcessPaths(
    SEXP src1, SEXP trg1, SEXP cl1, SEXP sg1, SEXP src1b, SEXP trg1b, SEXP cl1b, SEXP sg1b,
    SEXP src2, SEXP trg2, SEXP cl2, SEXP sg2, SEXP src3, SEXP trg3, SEXP cl3, SEXP sg3,
    SEXP src4, SEXP trg4, SEXP cl4, SEXP sg4, SEXP src5, SEXP trg5, SEXP cl5, SEXP sg5,
    SEXP inds1, SEXP inds1b, SEXP inds2, SEXP inds3, SEXP data1, SEXP data2, SEXP value_table,
    SEXP num_cases, SEXP num_ctrls, SEXP top_k, SEXP iterations, SEXP perm_cases, SEXP method, SEXP path_length,
    SEXP nthreads) {
  (void)src1; (void)src1b; (void)src2; (void)src3; (void)src4; (void)src5;
  int np = 0;   /* coerced copies, released together */
  load_abi();
  const int nc = Rf_asInteger(num_cases), nt = Rf_asInteger(num_ctrls), K = Rf_asInteger(iterations);
  /* "method1" -> 1, anything else -> 2 (JoinExec::to_method, gcre.h:125-133) */
  const int m = strcmp(CHAR(STRING_ELT(method, 0)), "method1") == 0 ? 1 : 2;

  gcre_pp_input in;
  memset(&in, 0, sizeof in);
  fill_level(&in.level[0], trg1, cl1, sg1, &np);
  fill_level(&in.level[1], trg1b, cl1b, sg1b, &np);
  fill_level(&in.level[2], trg2, cl2, sg2, &np);
  fill_level(&in.level[3], trg3, cl3, sg3, &np);
  fill_level(&in.level[4], trg4, cl4, sg4, &np);
  fill_level(&in.level[5], trg5, cl5, sg5, &np);
  SEXP inds[4] = {inds1, inds1b, inds2, inds3};
  for (int i = 0; i < 4; i++) {
    inds[i] = as_int(inds[i], &np);
    in.data_inds[i] = INTEGER(inds[i]);
    in.n_data_inds[i] = XLENGTH(inds[i]);
  }
  if (Rf_ncols(data1) != nc + nt || Rf_ncols(data2) != nc + nt)
    Rf_error("geneticsCRE: data matrices must have num_cases + num_ctrls columns");
  in.data1_rows = Rf_nrows(data1);
  in.data2_rows = Rf_nrows(data2);
  in.data1 = INTEGER(as_int(data1, &np));
  in.data2 = INTEGER(as_int(data2, &np));
  in.data_col_major = 1;                                /* R matrices are column-major (copy_r, wrapper.cpp:78-96) */
  in.vt_rows = Rf_nrows(value_table);
  in.vt_cols = Rf_ncols(value_table);
  in.value_table = REAL(as_real(value_table, &np));
  in.vt_col_major = 1;
  if (XLENGTH(perm_cases)) {                            /* matrix(0,0,0) when n_permutations == 0 */
    if (Rf_ncols(perm_cases) != nc + nt) Rf_error("geneticsCRE: perm_cases must have num_cases + num_ctrls columns");
    in.perm_rows = Rf_nrows(perm_cases);
    in.perm_cases = INTEGER(as_int(perm_cases, &np));
  }
  in.perm_col_major = 1;
  in.path_length = Rf_asInteger(path_length);

  /* Devices.  ONE by default (device 0): the several-GPU path (gcre_process_paths_devices with N > 1) has only been
   * rehearsed on a one-GPU box, so it is opt-in until it has run on a real node.  GCRE_DEVICES = "all" takes every GPU of
   * the node, a comma-separated list takes those ids.  `nthreads` -- the reference's worker count, wrapper.cpp:189 -- caps
   * the number of devices the call may take when it is positive (GWASPA(nthreads = 1) never takes a second GPU); R passes
   * -1 for nthreads = NA (ProcessPaths.R:106), which leaves the choice to GCRE_DEVICES.  The call creates and releases its
   * contexts itself: nothing native is alive when an R error (longjmp) can happen below. */
  int devs[64], ndev = 0;
  const int cap = Rf_asInteger(nthreads);
  const char* dl = getenv("GCRE_DEVICES");
  if (!dl || !*dl) {
    devs[ndev++] = 0;
  } else if (strcmp(dl, "all") == 0) {
    ndev = G.device_count();
    if (ndev > 64) ndev = 64;
    for (int i = 0; i < ndev; i++) devs[i] = i;
  } else {
    for (const char* q = dl; *q && ndev < 64;) {
      devs[ndev++] = atoi(q);
      q = strchr(q, ',');
      if (!q) break;
      q++;
    }
  }
  if (cap > 0 && ndev > cap) ndev = cap;
  gcre_result res[5];
  char msg[512] = "";
  const int rc = G.process_paths_devices(m, nc, nt, K, Rf_asInteger(top_k), ndev ? devs : NULL, ndev, &in, res, msg, sizeof msg);
  if (rc != GCRE_OK) Rf_error("geneticsCRE: %s", msg[0] ? msg : "gcre_process_paths_devices failed");

  /* all five lists are built before any result is freed, and the results are freed in one place -- on the error road too */
  SEXP out = R_ExecWithCleanup(build_result_list, res, release_results, res);
  UNPROTECT(np);
  return out;
}

/* getMatchingList (wrapper.cpp:60-69): named list uid -> c(count, location) */
SEXP _geneticsCRE_getMatchingList(SEXP uids, SEXP counts, SEXP location) {
  int np = 0;
  uids = as_int(uids, &np);
  counts = as_int(counts, &np);
  location = as_int(location, &np);
  const R_xlen_t n = XLENGTH(uids);
  SEXP out = PROTECT(Rf_allocVector(VECSXP, n));
  SEXP nms = PROTECT(Rf_allocVector(STRSXP, n));
  for (R_xlen_t i = 0; i < n; i++) {
    char key[32];
    SEXP cl = PROTECT(Rf_allocVector(INTSXP, 2));
    INTEGER(cl)[0] = INTEGER(counts)[i];
    INTEGER(cl)[1] = INTEGER(location)[i];
    SET_VECTOR_ELT(out, i, cl);
    snprintf(key, sizeof key, "%d", INTEGER(uids)[i]);
    SET_STRING_ELT(nms, i, Rf_mkChar(key));
    UNPROTECT(1);
  }
  Rf_setAttrib(out, R_NamesSymbol, nms);
  UNPROTECT(2 + np);
  return out;
}

/* getRels3 (wrapper.cpp:18-48): data.frame(srcuid, trguid, sign, trguid2, sign2), one row per 2-edge walk */
SEXP _geneticsCRE_getRels3(SEXP srcuid, SEXP trguid, SEXP sign, SEXP count_locs) {
  load_abi();
  int np = 0;
  srcuid = as_int(srcuid, &np);
  trguid = as_int(trguid, &np);
  sign = as_int(sign, &np);
  const R_xlen_t n = XLENGTH(trguid);
  int32_t *keys, *counts, *locs;
  R_xlen_t nk;
  flatten_count_locs(count_locs, &keys, &counts, &locs, &nk);
  int32_t* oc = (int32_t*)R_alloc(n ? n : 1, sizeof(int32_t));
  int64_t* ol = (int64_t*)R_alloc(n ? n : 1, sizeof(int64_t));
  if (G.resolve(INTEGER(trguid), n, keys, counts, locs, nk, oc, ol) != GCRE_OK) Rf_error("geneticsCRE: resolve failed");
  R_xlen_t total = 0;
  for (R_xlen_t i = 0; i < n; i++) total += oc[i] > 0 ? oc[i] : 0;
  SEXP cols[5];
  for (int c = 0; c < 5; c++) cols[c] = PROTECT(Rf_allocVector(INTSXP, total));
  R_xlen_t o = 0;
  for (R_xlen_t i = 0; i < n; i++)
    for (int64_t j = ol[i]; j < ol[i] + (oc[i] > 0 ? oc[i] : 0); j++, o++) {
      INTEGER(cols[0])[o] = INTEGER(srcuid)[i];
      INTEGER(cols[1])[o] = INTEGER(trguid)[i];
      INTEGER(cols[2])[o] = INTEGER(sign)[i];
      INTEGER(cols[3])[o] = INTEGER(trguid)[j];
      INTEGER(cols[4])[o] = INTEGER(sign)[j];
    }
  static const char* names[5] = {"srcuid", "trguid", "sign", "trguid2", "sign2"};
  SEXP df = PROTECT(Rf_allocVector(VECSXP, 5));
  SEXP nms = PROTECT(Rf_allocVector(STRSXP, 5));
  for (int c = 0; c < 5; c++) {
    SET_VECTOR_ELT(df, c, cols[c]);
    SET_STRING_ELT(nms, c, Rf_mkChar(names[c]));
  }
  Rf_setAttrib(df, R_NamesSymbol, nms);
  SEXP rn = PROTECT(Rf_allocVector(INTSXP, 2));         /* compact row names c(NA, -n) */
  INTEGER(rn)[0] = NA_INTEGER;
  INTEGER(rn)[1] = -(int)total;
  Rf_setAttrib(df, R_RowNamesSymbol, rn);
  Rf_setAttrib(df, R_ClassSymbol, Rf_mkString("data.frame"));
  UNPROTECT(8 + np);
  return df;
}

static const R_CallMethodDef CallEntries[] = {
    {"_geneticsCRE_getRels3", (DL_FUNC)&_geneticsCRE_getRels3, 4},
    {"_geneticsCRE_getMatchingList", (DL_FUNC)&_geneticsCRE_getMatchingList, 3},
    {"_geneticsCRE_ProcessPaths", (DL_FUNC)&_geneticsCRE_ProcessPaths, 39},
    {NULL, NULL, 0}};

void R_init_geneticsCRE(DllInfo* dll) {
  R_registerRoutines(dll, NULL, CallEntries, NULL, NULL);
  R_useDynamicSymbols(dll, FALSE);
}
