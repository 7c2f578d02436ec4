// TEMPORARY: per-cell stages not yet built into the library -- they fail loudly.
#include "lcs_internal.h"
extern "C" {
#define NOTYET(c) do { if (c) (c)->err = "per-cell stage not built into this library yet"; return LCS_ERR_BAD_ARG; } while (0)
int lcs_sss_detect(lcs_ctx *c, const lcs_cell *, const double *, uint32_t, double, double, double, double, lcs_cell *, double *, double *, double *, double *, double *, double *, double *, double *) { NOTYET(c); }
int lcs_pss_sss_foe(lcs_ctx *c, const lcs_cell *, const double *, uint32_t, double, double, double, lcs_cell *) { NOTYET(c); }
int lcs_extract_tfg(lcs_ctx *c, const lcs_cell *, const double *, uint32_t, double, double, double, double *, double *, int *) { NOTYET(c); }
int lcs_tfoec(lcs_ctx *c, const lcs_cell *, const double *, const double *, int, double, double, double *, double *, lcs_cell *) { NOTYET(c); }
int lcs_decode_mib(lcs_ctx *c, const lcs_cell *, const double *, int, lcs_cell *) { NOTYET(c); }
int lcs_search_capbuf(lcs_ctx *c, const double *, uint32_t, const double *, uint16_t, double, double, double, lcs_cell *, int, int *, lcs_cell *, int, int *) { NOTYET(c); }
}
