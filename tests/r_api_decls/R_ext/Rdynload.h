/* see ../Rinternals.h: declarations for a syntax check only, not R's header */
#pragma once
typedef void* (*DL_FUNC)(void);
typedef struct { const char* name; DL_FUNC fun; int numArgs; } R_CallMethodDef;
typedef struct _DllInfo DllInfo;
typedef enum { FALSE = 0, TRUE } Rboolean;
int R_registerRoutines(DllInfo*, const void*, const R_CallMethodDef*, const void*, const void*);
Rboolean R_useDynamicSymbols(DllInfo*, Rboolean);
