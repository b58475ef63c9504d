/* see Rinternals.h in this directory: declarations for a syntax check only, not R's header */
#pragma once
