/* pxo_solve.c -- ORACLE (test infrastructure only; see pxo.h header).  LM solvers: filled in below. */
#include "pxo.h"
