// CellSearch -- command-line front end of the MI355X-native searcher.
//
// Keeps the contract of the reference's CLI (src/CellSearch.cpp): the same options
// (-h -v -b -i -s -e -p -c -r -l -d, parse_commandline :92-280), the 100 kHz raster checks and
// warnings, the per-carrier progress lines, the "Detected a cell!" block, dedup (:285-319) and
// the final table (:575-614).  What differs: there is no RTL-SDR on a GPU node, so captures come
// from capbuf_NNNN.it files (-l, the reference's hardware-free mode, src/capbuf.cpp:98-115);
// without -l the program refuses to run.  With recorded data fc_programmed = fc_requested and fs_programmed = 1.92e6*correction
// (the convention of src/LTE-Tracker.cpp:609, 791; the reference's CellSearch leaves both
// uninitialised with -l, quirk Q5).  Extra option: -g/--gpu N selects the device.
//
// The searcher itself runs entirely on the GPU through include/searcher_amd.h -> include/lcs.h.
#include <getopt.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <list>
#include <sstream>
#include <string>
#include <vector>

#include "../include/searcher_amd.h"
#include "itfile.hpp"

using namespace std;
using lcs::Cell;

static int verbosity = 1;
#define VERSION_STRING "1.0.0-amd"

static void print_usage() {
  cout << "LTE CellSearch v" << VERSION_STRING << " (MI355X) help screen" << endl << endl;
  cout << "CellSearch -s start_frequency [optional_parameters]" << endl;
  cout << "  Basic options" << endl;
  cout << "    -h --help" << endl;
  cout << "      print this help screen" << endl;
  cout << "    -v --verbose" << endl;
  cout << "      increase status messages from program" << endl;
  cout << "    -b --brief" << endl;
  cout << "      reduce status messages from program" << endl;
  cout << "    -i --device-index N" << endl;
  cout << "      (accepted for compatibility; there is no RTLSDR dongle on a GPU node)" << endl;
  cout << "    -g --gpu N" << endl;
  cout << "      GPU to run the searcher on (default: current device)" << endl;
  cout << "  Frequency search options:" << endl;
  cout << "    -s --freq-start fs" << endl;
  cout << "      frequency where cell search should start" << endl;
  cout << "    -e --freq-end fe" << endl;
  cout << "      frequency where cell search should end" << endl;
  cout << "  Dongle LO correction options:" << endl;
  cout << "    -p --ppm ppm" << endl;
  cout << "      crystal remaining PPM error" << endl;
  cout << "    -c --correction c" << endl;
  cout << "      crystal correction factor" << endl;
  cout << "  Capture buffer save/ load options:" << endl;
  cout << "    -r --record" << endl;
  cout << "      save captured data in the files capbuf_XXXX.it" << endl;
  cout << "    -l --load" << endl;
  cout << "      used data in capbuf_XXXX.it files instead of live data" << endl;
  cout << "    -d --data-dir dir" << endl;
  cout << "      directory where capbuf_XXXX.it files are located" << endl << endl;
  cout << "'c' is the correction factor to apply and indicates that if the desired" << endl;
  cout << "center frequency is fc, the RTL-SDR dongle should be instructed to tune" << endl;
  cout << "to freqency fc*c so that its true frequency shall be fc. Default: 1.0" << endl << endl;
  cout << "'ppm' is the remaining frequency error of the crystal. Default: 120" << endl;
}

static double round_half_away(double x) { return (x < 0) ? -floor(-x + 0.5) : floor(x + 0.5); }   // itpp::round

static void parse_commandline(int argc, char *const argv[], double &freq_start, double &freq_end, double &ppm,
                              double &correction, bool &save_cap, bool &use_recorded_data, string &data_dir,
                              int &device_index, int &gpu) {
  freq_start = -1; freq_end = -1; ppm = 120; correction = 1; save_cap = false; use_recorded_data = false;
  data_dir = "."; device_index = -1; gpu = -1;
  static struct option long_options[] = {
      {"help", no_argument, 0, 'h'},          {"verbose", no_argument, 0, 'v'},     {"brief", no_argument, 0, 'b'},
      {"freq-start", required_argument, 0, 's'}, {"freq-end", required_argument, 0, 'e'}, {"ppm", required_argument, 0, 'p'},
      {"correction", required_argument, 0, 'c'}, {"record", no_argument, 0, 'r'},   {"load", no_argument, 0, 'l'},
      {"data-dir", required_argument, 0, 'd'},   {"device-index", required_argument, 0, 'i'}, {"gpu", required_argument, 0, 'g'},
      {0, 0, 0, 0}};
  while (1) {
    int option_index = 0;
    const int c = getopt_long(argc, argv, "hvbs:e:p:c:rld:i:g:", long_options, &option_index);
    if (c == -1) break;
    char *endp;
    switch (c) {
      case 'h': print_usage(); exit(-1);
      case 'v': verbosity = 2; break;
      case 'b': verbosity = 0; break;
      case 's':
        freq_start = strtod(optarg, &endp);
        if ((optarg == endp) || (*endp != '\0')) { cerr << "Error: could not parse start frequency" << endl; exit(-1); }
        break;
      case 'e':
        freq_end = strtod(optarg, &endp);
        if ((optarg == endp) || (*endp != '\0')) { cerr << "Error: could not parse end frequency" << endl; exit(-1); }
        break;
      case 'p':
        ppm = strtod(optarg, &endp);
        if ((optarg == endp) || (*endp != '\0')) { cerr << "Error: could not parse ppm value" << endl; exit(-1); }
        break;
      case 'c':
        correction = strtod(optarg, &endp);
        if ((optarg == endp) || (*endp != '\0')) { cerr << "Error: could not parse correction factor" << endl; exit(-1); }
        break;
      case 'r': save_cap = true; break;
      case 'l': use_recorded_data = true; break;
      case 'd': data_dir = optarg; break;
      case 'i':
        device_index = strtol(optarg, &endp, 10);
        if ((optarg == endp) || (*endp != '\0')) { cerr << "Error: could not parse device index" << endl; exit(-1); }
        if (device_index < 0) { cerr << "Error: device index cannot be negative" << endl; exit(-1); }
        break;
      case 'g':
        gpu = strtol(optarg, &endp, 10);
        if ((optarg == endp) || (*endp != '\0') || gpu < 0) { cerr << "Error: could not parse gpu index" << endl; exit(-1); }
        break;
      default: exit(-1);
    }
  }
  if (optind < argc) { cerr << "Error: unknown/extra arguments specified on command line" << endl; exit(-1); }
  if (freq_start == -1) { cerr << "Error: must specify a start frequency. (Try --help)" << endl; exit(-1); }
  if (freq_start < 1e6) { cerr << "Error: start frequency must be greater than 1MHz" << endl; exit(-1); }
  if (freq_start / 100e3 != round_half_away(freq_start / 100e3)) {
    freq_start = round_half_away(freq_start / 100e3) * 100e3;
    cout << "Warning: start frequency has been rounded to the nearest multiple of 100kHz" << endl;
  }
  if (freq_end == -1) freq_end = freq_start;
  if (freq_end < freq_start) { cerr << "Error: end frequency must be >= start frequency" << endl; exit(-1); }
  if (freq_end / 100e3 != round_half_away(freq_end / 100e3)) {
    freq_end = round_half_away(freq_end / 100e3) * 100e3;
    cout << "Warning: end frequency has been rounded to the nearest multiple of 100kHz" << endl;
  }
  if (ppm < 0) { cerr << "Error: ppm value must be positive" << endl; exit(-1); }
  if (ppm > 200) cout << "Warning: ppm value appears to be set unreasonably high" << endl;
  if (fabs(correction - 1) > 1000e-6) cout << "Warning: crystal correction factor appears to be unreasonable" << endl;
  if (save_cap && use_recorded_data) { cerr << "Error: cannot read and write captured data at the same time!" << endl; exit(-1); }
  if (verbosity >= 1) {
    cout << "LTE CellSearch v" << VERSION_STRING << " (MI355X) beginning" << endl;
    if (freq_start == freq_end) cout << "  Search frequency: " << freq_start / 1e6 << " MHz" << endl;
    else cout << "  Search frequency range: " << freq_start / 1e6 << "-" << freq_end / 1e6 << " MHz" << endl;
    cout << "  PPM: " << ppm << endl;
    stringstream temp;
    temp << setprecision(20) << correction;
    cout << "  correction: " << temp.str() << endl;
    if (use_recorded_data) cout << "  Captured data will be read from capbufXXXX.it files" << endl;
  }
}

// ref src/CellSearch.cpp:285-319
static void dedup(const vector<list<Cell> > &detected_cells, list<Cell> &cells_final) {
  cells_final.clear();
  for (size_t t = 0; t < detected_cells.size(); t++) {
    for (list<Cell>::const_iterator it_n = detected_cells[t].begin(); it_n != detected_cells[t].end(); ++it_n) {
      bool match = false;
      for (list<Cell>::iterator it_f = cells_final.begin(); it_f != cells_final.end(); ++it_f) {
        if ((it_n->n_id_cell() == it_f->n_id_cell()) &&
            (fabs((it_n->fc_requested + it_n->freq_superfine) - (it_f->fc_requested + it_f->freq_superfine)) < 1e6)) {
          match = true;
          if (it_n->pss_pow > it_f->pss_pow) *it_f = *it_n;
          break;
        }
      }
      if (!match) cells_final.push_back(*it_n);
    }
  }
}

// ref src/CellSearch.cpp:322-340
static string freq_formatter(double freq) {
  stringstream temp;
  if (fabs(freq) < 998.0) temp << setw(5) << setprecision(3) << freq << "h";
  else if (fabs(freq) < 998000.0) temp << setw(5) << setprecision(3) << freq / 1e3 << "k";
  else if (fabs(freq) < 998000000.0) temp << setw(5) << setprecision(3) << freq / 1e6 << "m";
  else if (fabs(freq) < 998000000000.0) temp << setw(5) << setprecision(3) << freq / 1e9 << "g";
  else if (fabs(freq) < 998000000000000.0) temp << setw(5) << setprecision(3) << freq / 1e12 << "t";
  else temp << freq;
  return temp.str();
}

static double db10(double s) { return 10 * log10(s); }

int main(int argc, char *const argv[]) {
  double freq_start, freq_end, ppm, correction;
  bool save_cap, use_recorded_data;
  string data_dir;
  int device_index, gpu;
  parse_commandline(argc, argv, freq_start, freq_end, ppm, correction, save_cap, use_recorded_data, data_dir, device_index, gpu);
  if (!use_recorded_data) {
    cerr << "Error: this build has no RTL-SDR support (GPU node); use --load with capbuf_XXXX.it files" << endl;
    return 1;
  }
  const double fs_programmed = 1.92e6 * correction;   // recorded-data convention, src/LTE-Tracker.cpp:791

  // frequency-offset and carrier grids (src/CellSearch.cpp:463-465)
  const int n_extra = (int)floor((freq_start * ppm / 1e6 + 2.5e3) / 5e3);
  lcsc::vec f_search_set(2 * n_extra + 1);
  for (int i = 0; i < 2 * n_extra + 1; ++i) f_search_set(i) = -n_extra * 5000.0 + 5000.0 * i;
  const int n_fc = (int)floor((freq_end - freq_start) / 100e3) + 1;

  lcs::Searcher *searcher = 0;
  try {
    searcher = new lcs::Searcher(gpu);
  } catch (const std::exception &e) {
    cerr << "Error: " << e.what() << endl;
    return 2;
  }

  vector<list<Cell> > detected_cells(n_fc);
  for (int fci = 0; fci < n_fc; fci++) {
    const double fc_requested = freq_start + 100e3 * fci;
    if (verbosity >= 1) cout << "Examining center frequency " << fc_requested / 1e6 << " MHz ..." << endl;
    stringstream filename;
    filename << data_dir << "/capbuf_" << setw(4) << setfill('0') << fci << ".it";
    if (verbosity >= 2) cout << "Reading captured data from file: " << filename.str() << endl;
    lcsc::cvec capbuf;
    try {
      map<string, itfile::Var> vars = itfile::read_all(filename.str());
      vector<complex<double> > cb = itfile::get_dcvec(vars, "capbuf");
      vector<int32_t> fc_v = itfile::get_ivec(vars, "fc");
      capbuf.set_size((int)cb.size());
      for (size_t i = 0; i < cb.size(); ++i) capbuf((int)i) = cb[i];
      if (fc_v.empty() || fc_requested != fc_v[0]) {
        cout << "Warning: while reading capture buffer " << fci << ", the read" << endl;
        cout << "center frequency did not match the expected center frequency." << endl;
      }
    } catch (const std::exception &e) {
      cerr << "Error: " << e.what() << endl;
      delete searcher;
      return 3;
    }
    const double fc_programmed = fc_requested;
    if (verbosity >= 2) cout << "  Calculating PSS correlations and examining correlation peaks (GPU)..." << endl;
    try {
      searcher->search_capbuf(capbuf, f_search_set, fc_requested, fc_programmed, fs_programmed, detected_cells[fci]);
    } catch (const std::exception &e) {
      cerr << "Error: " << e.what() << endl;
      delete searcher;
      return 4;
    }
    if (verbosity >= 1) {
      for (list<Cell>::iterator it = detected_cells[fci].begin(); it != detected_cells[fci].end(); ++it) {
        cout << "  Detected a cell!" << endl;
        cout << "    cell ID: " << it->n_id_cell() << endl;
        cout << "    RX power level: " << db10(it->pss_pow) << " dB" << endl;
        cout << "    residual frequency offset: " << it->freq_superfine << " Hz" << endl;
      }
    }
  }
  delete searcher;

  list<Cell> cells_final;
  dedup(detected_cells, cells_final);
  if (cells_final.size() == 0) {
    cout << "No LTE cells were found..." << endl;
  } else {
    cout << "Detected the following cells:" << endl;
    cout << "A: #antenna ports C: CP type ; P: PHICH duration ; PR: PHICH resource type" << endl;
    cout << "CID A      fc   foff RXPWR C nRB P  PR CrystalCorrectionFactor" << endl;
    for (list<Cell>::iterator it = cells_final.begin(); it != cells_final.end(); ++it) {
      stringstream ss;
      ss << setw(3) << it->n_id_cell();
      ss << setw(2) << it->n_ports;
      ss << " " << setw(6) << setprecision(5) << it->fc_requested / 1e6 << "M";
      ss << " " << freq_formatter(it->freq_superfine);
      ss << " " << setw(5) << setprecision(3) << db10(it->pss_pow);
      ss << " " << ((it->cp_type == LCS_CP_NORMAL) ? "N" : ((it->cp_type == LCS_CP_UNKNOWN) ? "U" : "E"));
      ss << " " << setw(3) << it->n_rb_dl;
      ss << " " << ((it->phich_duration == 1) ? "N" : ((it->phich_duration == 0) ? "U" : "E"));
      switch (it->phich_resource) {
        case 0: ss << " UNK"; break;
        case 1: ss << " 1/6"; break;
        case 2: ss << " 1/2"; break;
        case 3: ss << " one"; break;
        case 4: ss << " two"; break;
      }
      const double true_location = it->fc_requested;
      const double crystal_freq_actual = it->fc_requested - it->freq_superfine;
      const double correction_residual = true_location / crystal_freq_actual;
      const double correction_new = correction * correction_residual;
      ss << " " << setprecision(20) << correction_new;
      cout << ss.str() << endl;
    }
  }
  return 0;
}
