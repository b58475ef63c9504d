/* tests/r_api_decls/Rinternals.h -- NOT R's header.  Declarations of the few R C-API entry points
 * geneticscre_amd/csrc/r_shim.c uses, with the signatures documented in "Writing R Extensions" (sections 5.9, 5.10),
 * so that the shim can be syntax- and type-checked (gcc -fsyntax-only) in an image without R.  Test
 * infrastructure only: nothing is built or linked against this file.  On a machine with R the real headers are used. */
#pragma once
#include <stddef.h>
typedef struct SEXPREC* SEXP;
typedef ptrdiff_t R_xlen_t;
typedef unsigned int SEXPTYPE;
#define INTSXP 13
#define REALSXP 14
#define STRSXP 16
#define VECSXP 19
extern SEXP R_NamesSymbol, R_RowNamesSymbol, R_ClassSymbol;
extern int R_NaInt;
#define NA_INTEGER R_NaInt
SEXP Rf_protect(SEXP);
void Rf_unprotect(int);
#define PROTECT(s) Rf_protect(s)
#define UNPROTECT(n) Rf_unprotect(n)
int TYPEOF(SEXP);
R_xlen_t XLENGTH(SEXP);
int* INTEGER(SEXP);
double* REAL(SEXP);
const char* R_CHAR(SEXP);
#define CHAR(x) R_CHAR(x)
SEXP VECTOR_ELT(SEXP, R_xlen_t);
SEXP SET_VECTOR_ELT(SEXP, R_xlen_t, SEXP);
SEXP STRING_ELT(SEXP, R_xlen_t);
void SET_STRING_ELT(SEXP, R_xlen_t, SEXP);
SEXP Rf_allocVector(SEXPTYPE, R_xlen_t);
SEXP Rf_allocMatrix(SEXPTYPE, int, int);
SEXP Rf_coerceVector(SEXP, SEXPTYPE);
SEXP Rf_getAttrib(SEXP, SEXP);
SEXP Rf_setAttrib(SEXP, SEXP, SEXP);
SEXP Rf_mkChar(const char*);
SEXP Rf_mkString(const char*);
int Rf_asInteger(SEXP);
double Rf_asReal(SEXP);
int Rf_nrows(SEXP);
int Rf_ncols(SEXP);
void Rf_error(const char*, ...) __attribute__((noreturn));
char* R_alloc(size_t, int);
SEXP R_ExecWithCleanup(SEXP (*fun)(void*), void* data, void (*cleanfun)(void*), void* cleandata);
