// searcher_shim.h -- the reference's searcher prototypes (include/searcher.h:22-124), name for name and
// argument for argument, implemented by host/searcher_shim.cpp on top of liblcs_amd.so.
//
// In the reference tree this header is NOT needed: searcher_shim.cpp is compiled there with
// -DLCS_SHIM_REFERENCE_TREE against the reference's own <itpp/itbase.h>, common.h, lte_lib.h and searcher.h and
// replaces src/searcher.cpp in libLTE_MISC.a.  IT++ is not installed where this
// repository is built, so here the same source is compiled against the stand-ins below: the container types of
// include/lcs_containers.h under the name `itpp`, and `Cell` / `RS_DL` / the enums with the reference's member
// names and types (include/common.h.in:41-129, include/lte_lib.h:68-88) -- enough for the reference's call sites
// (src/CellSearch.cpp:484-558) to compile unchanged against it, which host/test_shim.cpp does.
#ifndef LCS_SEARCHER_SHIM_H
#define LCS_SEARCHER_SHIM_H

#ifdef LCS_SHIM_REFERENCE_TREE
#include <itpp/itbase.h>
#include <list>
#include "common.h"
#include "lte_lib.h"
#include "searcher.h"
#else
#include <stdint.h>
#include <cmath>
#include <complex>
#include <list>
#include <vector>
#include "../include/lcs_containers.h"

namespace itpp = lcsc;                         // stand-in containers under the reference's namespace name
typedef uint8_t uint8;                         // itpp/base/ittypes.h
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef int8_t int8;
typedef int16_t int16;
typedef int32_t int32;

// include/common.h.in:41-44
typedef std::vector<std::vector<std::vector<std::complex<float> > > > vcf3d;
typedef std::vector<std::vector<std::vector<float> > > vf3d;
// include/common.h.in:48-96
namespace cp_type_t { enum cp_type_t { UNKNOWN = 0, NORMAL, EXTENDED }; }
namespace phich_duration_t { enum phich_duration_t { UNKNOWN = 0, NORMAL, EXTENDED }; }
namespace phich_resource_t { enum phich_resource_t { UNKNOWN = 0, oneSixth, half, one, two }; }

// include/common.h.in:101-129, src/common.cpp:29-56
class Cell {
 public:
  double fc_requested;
  double fc_programmed;
  double pss_pow;
  int32 ind;
  double freq;
  int8 n_id_2;
  int16 n_id_1;
  cp_type_t::cp_type_t cp_type;
  double frame_start;
  double freq_fine;
  double freq_superfine;
  int8 n_ports;
  int8 n_rb_dl;
  phich_duration_t::phich_duration_t phich_duration;
  phich_resource_t::phich_resource_t phich_resource;
  int16 sfn;
  Cell()
      : fc_requested(NAN), fc_programmed(NAN), pss_pow(NAN), ind(-1), freq(NAN), n_id_2(-1), n_id_1(-1),
        cp_type(cp_type_t::UNKNOWN), frame_start(NAN), freq_fine(NAN), freq_superfine(NAN), n_ports(-1), n_rb_dl(-1),
        phich_duration(phich_duration_t::UNKNOWN), phich_resource(phich_resource_t::UNKNOWN), sfn(-1) {}
  int16 const n_id_cell() const { return n_id_2 + 3 * n_id_1; }
  int8 const n_symb_dl() const { return (cp_type == cp_type_t::NORMAL) ? 7 : ((cp_type == cp_type_t::EXTENDED) ? 6 : -1); }
};

// include/lte_lib.h:68-88.  The reference tabulates the cell-specific reference signals on the host and hands the
// table to tfoec / decode_mib; liblcs_amd rebuilds it on the device from (n_id_cell, cp_type), so the stand-in only
// records its constructor arguments.
class RS_DL {
 public:
  RS_DL(const uint16 &n_id_cell, const uint8 &n_rb_dl, const cp_type_t::cp_type_t &cp_type)
      : n_id_cell_(n_id_cell), n_rb_dl_(n_rb_dl), cp_type_(cp_type) {}
  uint16 n_id_cell_;
  uint8 n_rb_dl_;
  cp_type_t::cp_type_t cp_type_;
};

// ---- include/searcher.h:22-124 ------------------------------------------------------------------------------
void xcorr_pss(const itpp::cvec &capbuf, const itpp::vec &f_search_set, const uint8 &ds_comb_arm,
               const double &fc_requested, const double &fc_programmed, const double &fs_programmed,
               itpp::mat &xc_incoherent_collapsed_pow, itpp::imat &xc_incoherent_collapsed_frq,
               vf3d &xc_incoherent_single, vf3d &xc_incoherent, itpp::vec &sp_incoherent, vcf3d &xc, itpp::vec &sp,
               uint16 &n_comb_xc, uint16 &n_comb_sp);
void peak_search(const itpp::mat &xc_incoherent_collapsed_pow, const itpp::imat &xc_incoherent_collapsed_frq,
                 const itpp::vec &Z_th1, const itpp::vec &f_search_set, const double &fc_requested,
                 const double &fc_programmed, const vf3d &xc_incoherent_single, const uint8 &ds_comb_arm,
                 std::list<Cell> &cells);
Cell sss_detect(const Cell &cell, const itpp::cvec &capbuf, const double &thresh2_n_sigma, const double &fc_requested,
                const double &fc_programmed, const double &fs_programmed, itpp::vec &sss_h1_np_est,
                itpp::vec &sss_h2_np_est, itpp::cvec &sss_h1_nrm_est, itpp::cvec &sss_h2_nrm_est,
                itpp::cvec &sss_h1_ext_est, itpp::cvec &sss_h2_ext_est, itpp::mat &log_lik_nrm, itpp::mat &log_lik_ext);
Cell pss_sss_foe(const Cell &cell_in, const itpp::cvec &capbuf, const double &fc_requested, const double &fc_programmed,
                 const double &fs_programmed);
void extract_tfg(const Cell &cell, const itpp::cvec &capbuf_raw, const double &fc_requested, const double &fc_programmed,
                 const double &fs_programmed, itpp::cmat &tfg, itpp::vec &tfg_timestamp);
Cell tfoec(const Cell &cell, const itpp::cmat &tfg, const itpp::vec &tfg_timestamp, const double &fc_requested,
           const double &fc_programmed, const RS_DL &rs_dl, itpp::cmat &tfg_comp, itpp::vec &tfg_comp_timestamp);
Cell decode_mib(const Cell &cell, const itpp::cmat &tfg, const RS_DL &rs_dl);
void del_oob(itpp::ivec &v);
#endif  // LCS_SHIM_REFERENCE_TREE

// Which GPU the shim's per-thread contexts use (default: the current HIP device).  Not part of the reference's API.
void lcs_shim_set_device(int device);
// xcorr_pss fills its `vcf3d &xc` argument (the raw correlations, a debug output: include/searcher.h:35,
// test/test_xcorr_pss.cpp:104-109) only after lcs_shim_want_xc(true); default off, `xc` then comes back empty
void lcs_shim_want_xc(bool on);
#endif
