// CellSearch -- command-line front end of the MI355X-native searcher.
//
// Contract kept from the reference's CLI (src/CellSearch.cpp): option letters and long names (:43-87, :117-133), the
// 100 kHz raster rounding and its warnings (:222-258), the banner, the per-carrier progress line, the
// "Detected a cell!" block, the de-duplication rule (:285-319) and the result table (:575-614) -- byte for byte on
// stdout.  What is NOT kept is the program's shape: options are a table walked by a small scanner, captures are read
// for the whole sweep and pushed through the GPU in batches, cells are de-duplicated through an index by cell
// identity, and every number is formatted by one helper.
//
// There is no RTL-SDR on a GPU node: captures come from capbuf_NNNN.it files (-l, the reference's hardware-free
// mode, src/capbuf.cpp:98-115); without -l the program refuses to run.  With recorded data fc_programmed =
// fc_requested and fs_programmed = 1.92e6 * correction (the convention of src/LTE-Tracker.cpp:609, 791).
// A recorded capture holds exactly (u8-127)/128 per component (src/capbuf.cpp:172-181): such buffers go to the GPU
// as raw bytes, 64 carriers per batch, and take the int8 correlation kernel; anything else (synthetic complex
// data) is searched one buffer at a time as complex<double>.  Extra option: -g/--gpu N selects the device, -g all
// shards the carrier sweep over every visible GPU (one thread and two batches in flight per device), -g a,b,... over the
// listed devices (an index may repeat).
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <iostream>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/searcher_amd.h"
#include "itfile.hpp"

using lcs::Cell;

#define VERSION_STRING "1.0.0-amd"

namespace {

// ---- formatting: iostream's default float notation with precision p and width w is printf's %*.{p}g ----------
std::string fmt(const char *f, ...) {
  char buf[256];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof(buf), f, ap);
  va_end(ap);
  return buf;
}
std::string num(double v, int precision = 6, int width = 0) { return fmt("%*.*g", width, precision, v); }

// ---- options ---------------------------------------------------------------------------------------------------
struct Options {
  double freq_start = -1, freq_end = -1, ppm = 120, correction = 1;
  bool record = false, load = false, help = false;
  std::string data_dir = ".";
  long device_index = -1, gpu = -1;
  bool gpu_all = false;          // -g all: shard the carriers over every visible GPU
  std::vector<long> gpu_list;    // -g a,b,...: one device thread per entry (an index may repeat: "0,0" = two threads on GPU 0)
  long batch = 64;               // carriers per GPU batch (one correlation launch)
  int verbosity = 1;
};

enum ArgKind { FLAG, REAL, INDEX, TEXT };
struct OptSpec {
  char letter;
  const char *name;
  ArgKind kind;
  const char *what;       // "could not parse <what>"
  const char *help[2];    // usage lines
};
const OptSpec kSpecs[] = {
    {'h', "help", FLAG, 0, {"print this help screen", 0}},
    {'v', "verbose", FLAG, 0, {"increase status messages from program", 0}},
    {'b', "brief", FLAG, 0, {"reduce status messages from program", 0}},
    {'i', "device-index", INDEX, "device index", {"(accepted for compatibility; there is no RTLSDR dongle on a GPU node)", 0}},
    {'g', "gpu", INDEX, "gpu index", {"GPU to run the searcher on (default: current device); 'all' shards the carriers over every GPU, 'a,b,..' over the listed ones", 0}},
    {'B', "batch", INDEX, "batch size", {"carriers searched per GPU batch (default 64)", 0}},
    {'s', "freq-start", REAL, "start frequency", {"frequency where cell search should start", 0}},
    {'e', "freq-end", REAL, "end frequency", {"frequency where cell search should end", 0}},
    {'p', "ppm", REAL, "ppm value", {"crystal remaining PPM error", 0}},
    {'c', "correction", REAL, "correction factor", {"crystal correction factor", 0}},
    {'r', "record", FLAG, 0, {"save captured data in the files capbuf_XXXX.it", 0}},
    {'l', "load", FLAG, 0, {"used data in capbuf_XXXX.it files instead of live data", 0}},
    {'d', "data-dir", TEXT, 0, {"directory where capbuf_XXXX.it files are located", 0}},
};
const char *arg_name(const OptSpec &s) { return s.kind == FLAG ? "" : (s.letter == 's' ? " fs" : s.letter == 'e' ? " fe" : s.letter == 'p' ? " ppm" : s.letter == 'c' ? " c" : s.letter == 'd' ? " dir" : " N"); }

void usage() {
  std::cout << "LTE CellSearch v" << VERSION_STRING << " (MI355X) help screen\n\n"
            << "CellSearch -s start_frequency [optional_parameters]\n";
  const struct { const char *title; const char *letters; } sections[] = {
      {"Basic options", "hvbigB"}, {"Frequency search options:", "se"}, {"Dongle LO correction options:", "pc"},
      {"Capture buffer save/ load options:", "rld"}};
  for (const auto &sec : sections) {
    std::cout << "  " << sec.title << "\n";
    for (const char *l = sec.letters; *l; ++l)
      for (const OptSpec &s : kSpecs)
        if (s.letter == *l) std::cout << "    -" << s.letter << " --" << s.name << arg_name(s) << "\n      " << s.help[0] << "\n";
  }
  std::cout << "\n'c' is the correction factor to apply and indicates that if the desired\n"
            << "center frequency is fc, the RTL-SDR dongle should be instructed to tune\n"
            << "to freqency fc*c so that its true frequency shall be fc. Default: 1.0\n\n"
            << "'ppm' is the remaining frequency error of the crystal. Default: 120\n";
}

[[noreturn]] void die(const std::string &msg) {
  std::cerr << "Error: " << msg << std::endl;
  std::exit(-1);
}

void store(Options &o, const OptSpec &s, const char *value) {
  char *end = 0;
  switch (s.kind) {
    case FLAG:
      if (s.letter == 'h') o.help = true;
      else if (s.letter == 'v') o.verbosity = 2;
      else if (s.letter == 'b') o.verbosity = 0;
      else if (s.letter == 'r') o.record = true;
      else o.load = true;
      return;
    case TEXT: o.data_dir = value; return;
    case REAL: {
      const double v = std::strtod(value, &end);
      if (end == value || *end) die(std::string("could not parse ") + s.what);
      (s.letter == 's' ? o.freq_start : s.letter == 'e' ? o.freq_end : s.letter == 'p' ? o.ppm : o.correction) = v;
      return;
    }
    case INDEX: {
      if (s.letter == 'g' && std::strcmp(value, "all") == 0) { o.gpu_all = true; return; }
      if (s.letter == 'g' && std::strchr(value, ',')) {       // device list
        o.gpu_list.clear();
        for (const char *p = value;;) {
          const long v = std::strtol(p, &end, 10);
          if (end == p || v < 0 || (*end && *end != ',')) die("could not parse gpu index");
          o.gpu_list.push_back(v);
          if (!*end) break;
          p = end + 1;
        }
        return;
      }
      const long v = std::strtol(value, &end, 10);
      if (end == value || *end) die(std::string("could not parse ") + s.what);
      if (v < 0) die(s.letter == 'i' ? "device index cannot be negative" : s.letter == 'g' ? "could not parse gpu index" : "could not parse batch size");
      if (s.letter == 'B') { if (v < 1 || v > 512) die("batch size must be 1..512"); o.batch = v; return; }
      (s.letter == 'i' ? o.device_index : o.gpu) = v;
      return;
    }
  }
}

// -x, -xVALUE, -x VALUE, clustered flags (-vl), --name, --name VALUE, --name=VALUE; anything else is "extra"
Options scan_args(int argc, char *const argv[]) {
  Options o;
  for (int i = 1; i < argc; ++i) {
    const char *a = argv[i];
    if (a[0] != '-' || !a[1]) die("unknown/extra arguments specified on command line");
    if (a[1] == '-') {
      const char *eq = std::strchr(a + 2, '=');
      const std::string name = eq ? std::string(a + 2, eq) : std::string(a + 2);
      const OptSpec *hit = 0;
      for (const OptSpec &s : kSpecs) if (name == s.name) hit = &s;
      if (!hit) std::exit(-1);                                   // unknown option: the reference exits silently after getopt's message
      if (hit->kind == FLAG) { store(o, *hit, 0); continue; }
      if (!eq && i + 1 >= argc) std::exit(-1);
      store(o, *hit, eq ? eq + 1 : argv[++i]);
      continue;
    }
    for (const char *p = a + 1; *p; ++p) {
      const OptSpec *hit = 0;
      for (const OptSpec &s : kSpecs) if (*p == s.letter) hit = &s;
      if (!hit) std::exit(-1);
      if (hit->kind == FLAG) { store(o, *hit, 0); continue; }
      if (p[1]) { store(o, *hit, p + 1); break; }
      if (i + 1 >= argc) std::exit(-1);
      store(o, *hit, argv[++i]);
      break;
    }
  }
  if (o.help) { usage(); std::exit(-1); }
  return o;
}

double nearest_raster(double f) { return (f < 0 ? -std::floor(-f / 100e3 + 0.5) : std::floor(f / 100e3 + 0.5)) * 100e3; }   // itpp::round

void validate(Options &o) {
  if (o.freq_start == -1) die("must specify a start frequency. (Try --help)");
  if (o.freq_start < 1e6) die("start frequency must be greater than 1MHz");
  struct { double *f; const char *which; } ends[2] = {{&o.freq_start, "start"}, {&o.freq_end, "end"}};
  for (int k = 0; k < 2; ++k) {
    if (k == 1) {
      if (o.freq_end == -1) o.freq_end = o.freq_start;
      if (o.freq_end < o.freq_start) die("end frequency must be >= start frequency");
    }
    if (nearest_raster(*ends[k].f) != *ends[k].f) {
      *ends[k].f = nearest_raster(*ends[k].f);
      std::cout << "Warning: " << ends[k].which << " frequency has been rounded to the nearest multiple of 100kHz" << std::endl;
    }
  }
  if (o.ppm < 0) die("ppm value must be positive");
  if (o.ppm > 200) std::cout << "Warning: ppm value appears to be set unreasonably high" << std::endl;
  if (std::fabs(o.correction - 1) > 1000e-6) std::cout << "Warning: crystal correction factor appears to be unreasonable" << std::endl;
  if (o.record && o.load) die("cannot read and write captured data at the same time!");
  if (o.verbosity >= 1) {
    std::cout << "LTE CellSearch v" << VERSION_STRING << " (MI355X) beginning\n";
    if (o.freq_start == o.freq_end) std::cout << "  Search frequency: " << num(o.freq_start / 1e6) << " MHz\n";
    else std::cout << "  Search frequency range: " << num(o.freq_start / 1e6) << "-" << num(o.freq_end / 1e6) << " MHz\n";
    std::cout << "  PPM: " << num(o.ppm) << "\n  correction: " << num(o.correction, 20) << "\n";
    if (o.load) std::cout << "  Captured data will be read from capbufXXXX.it files\n";
    std::cout.flush();
  }
}

// ---- captures ------------------------------------------------------------------------------------------------------
struct Capture {
  std::vector<std::complex<double> > samples;
  std::vector<unsigned char> iq_u8;      // filled when every component is exactly (u8-127)/128
  bool fc_matches = true;
};

Capture read_capture(const std::string &path, double fc_expected) {
  Capture c;
  std::map<std::string, itfile::Var> vars = itfile::read_all(path);
  c.samples = itfile::get_dcvec(vars, "capbuf");
  const std::vector<int32_t> fc = itfile::get_ivec(vars, "fc");
  c.fc_matches = !fc.empty() && fc_expected == fc[0];
  c.iq_u8.resize(2 * c.samples.size());
  const double *x = reinterpret_cast<const double *>(c.samples.data());
  for (size_t i = 0; i < c.iq_u8.size(); ++i) {
    const double code = x[i] * 128.0 + 127.0;               // exact for dongle data
    if (!(code >= 0.0 && code <= 255.0) || code != std::floor(code)) { c.iq_u8.clear(); break; }
    c.iq_u8[i] = (unsigned char)code;
  }
  return c;
}

// ---- results ---------------------------------------------------------------------------------------------------------
// The reference's rule (src/CellSearch.cpp:285-319): walk the carriers in order; a cell is the same as an earlier one
// when the identity matches and the two true centre frequencies lie within 1 MHz; of the two, the stronger PSS
// stays, in the earlier one's place in the list.
std::vector<Cell> merge_duplicates(const std::vector<std::list<Cell> > &per_carrier) {
  std::vector<Cell> kept;
  std::multimap<int, size_t> by_identity;                    // n_id_cell -> positions in `kept`, in insertion order
  for (const std::list<Cell> &found : per_carrier)
    for (const Cell &c : found) {
      const double f_true = c.fc_requested + c.freq_superfine;
      size_t twin = kept.size();
      const auto range = by_identity.equal_range(c.n_id_cell());
      for (auto it = range.first; it != range.second && twin == kept.size(); ++it)
        if (std::fabs(f_true - (kept[it->second].fc_requested + kept[it->second].freq_superfine)) < 1e6) twin = it->second;
      if (twin == kept.size()) {
        by_identity.insert(std::make_pair(c.n_id_cell(), kept.size()));
        kept.push_back(c);
      } else if (c.pss_pow > kept[twin].pss_pow) {
        kept[twin] = c;
      }
    }
  return kept;
}

// 12.3k / 35.2k / 1.02m ...: three significant digits and an SI letter (src/CellSearch.cpp:322-340)
std::string si_frequency(double f) {
  static const struct { double below, unit; const char *suffix; } steps[] = {
      {998.0, 1.0, "h"}, {998e3, 1e3, "k"}, {998e6, 1e6, "m"}, {998e9, 1e9, "g"}, {998e12, 1e12, "t"}};
  for (const auto &s : steps)
    if (std::fabs(f) < s.below) return num(f / s.unit, 3, 5) + s.suffix;
  return num(f);
}

double db10(double p) { return 10 * std::log10(p); }

std::string table_row(const Cell &c, double correction) {
  static const char *cp[] = {"U", "N", "E"}, *pd[] = {"U", "N", "E"}, *pr[] = {" UNK", " 1/6", " 1/2", " one", " two"};
  const double crystal_freq_actual = c.fc_requested - c.freq_superfine;       // :601-605
  const double correction_new = correction * (c.fc_requested / crystal_freq_actual);
  return fmt("%3d%2d ", c.n_id_cell(), c.n_ports) + num(c.fc_requested / 1e6, 5, 6) + "M " + si_frequency(c.freq_superfine) + " " +
         num(db10(c.pss_pow), 3, 5) + " " + cp[c.cp_type] + fmt(" %3d ", c.n_rb_dl) + pd[c.phich_duration] +
         (c.phich_resource >= 0 && c.phich_resource <= 4 ? pr[c.phich_resource] : "") + " " + num(correction_new, 20);
}

void announce(const std::list<Cell> &cells) {
  for (const Cell &c : cells)
    std::cout << "  Detected a cell!\n    cell ID: " << c.n_id_cell() << "\n    RX power level: " << num(db10(c.pss_pow))
              << " dB\n    residual frequency offset: " << num(c.freq_superfine) << " Hz" << std::endl;
}

// ---- the sweep -------------------------------------------------------------------------------------------------------
// The carrier loop of src/CellSearch.cpp:471-569, sharded: carriers are cut into batches of 64 (one correlation launch),
// batch b belongs to GPU b mod n_gpus (block-cyclic), every GPU has one thread that keeps TWO batches in flight -- it
// reads and enqueues batch k + 1 (page-locked staging buffer, asynchronous copy) while the device works on batch k -- and
// the main thread prints the per-carrier report in carrier order as batches complete.  No collective is needed: the
// results meet in host memory of this one process.
struct Sweep {
  Options opt;
  lcsc::vec f_search_set;
  double fs_programmed;
  int n_fc, n_batches;
  std::vector<std::list<Cell> > detected;
  std::vector<char> fc_matches, batched;
  std::vector<char> batch_done;
  std::string failure;
  std::mutex m;
  std::condition_variable cv;
  int kBatch = 64;                     // carriers per batch (-B)

  void finish(int b) {
    std::lock_guard<std::mutex> lk(m);
    batch_done[b] = 1;
    cv.notify_all();
  }
  void fail(const std::string &what) {
    std::lock_guard<std::mutex> lk(m);
    if (failure.empty()) failure = what;
    cv.notify_all();
  }
};

struct InFlight {
  int batch = -1;
  std::vector<int> carriers;           // carriers of the batch that went to the GPU as bytes, in buffer order
  std::vector<Capture> singles;        // the others: searched one by one as complex<double>
  std::vector<int> single_carriers;
};

void device_thread(Sweep *sw, int device, int first_batch, int stride) {
  try {
    std::unique_ptr<lcs::Searcher> ctx[2];
    for (int k = 0; k < 2; ++k) ctx[k].reset(new lcs::Searcher(device));
    std::unique_ptr<lcs::Searcher> one;                     // for captures that are not raw dongle bytes
    unsigned char *pinned[2] = {0, 0};
    size_t pinned_bytes[2] = {0, 0};
    InFlight fl[2];
    auto collect = [&](int slot) {
      InFlight &f = fl[slot];
      if (f.batch < 0) return;
      if (!f.carriers.empty()) {
        std::vector<std::list<Cell> > found;
        ctx[slot]->collect_batch(found);
        if (ctx[slot]->last_batch_overflowed()) std::cerr << "Warning: more cells than the result arrays hold; list truncated" << std::endl;
        for (size_t j = 0; j < f.carriers.size(); ++j) sw->detected[f.carriers[j]].swap(found[j]);
      }
      for (size_t j = 0; j < f.singles.size(); ++j) {
        if (!one) one.reset(new lcs::Searcher(device));
        const double fc = sw->opt.freq_start + 100e3 * f.single_carriers[j];
        lcsc::cvec capbuf((int)f.singles[j].samples.size());
        std::memcpy(capbuf._data(), f.singles[j].samples.data(), f.singles[j].samples.size() * sizeof(std::complex<double>));
        one->search_capbuf(capbuf, sw->f_search_set, fc, fc, sw->fs_programmed, sw->detected[f.single_carriers[j]]);
      }
      sw->finish(f.batch);
      f = InFlight();
    };
    int slot = 0;
    for (int b = first_batch; b < sw->n_batches; b += stride, slot ^= 1) {
      collect(slot);                                        // the batch this slot carried two rounds ago
      const int first = b * sw->kBatch, n = std::min<int>(sw->kBatch, sw->n_fc - first);
      InFlight &f = fl[slot];
      f.batch = b;
      // a batch's files are read and converted by a few threads at once: parsing 64 x 2.46 MB of complex<double> is what a
      // sweep over recorded captures spends its time on (the GPU needs ~1.5 ms for the batch)
      std::vector<Capture> caps(n);
      {
        std::atomic<int> next(0);
        std::string read_error;
        auto reader = [&]() {
          for (int k = next.fetch_add(1); k < n; k = next.fetch_add(1)) {
            const std::string path = sw->opt.data_dir + fmt("/capbuf_%04d.it", first + k);
            if (sw->opt.verbosity >= 2) { std::lock_guard<std::mutex> lk(sw->m); std::cout << "Reading captured data from file: " << path << std::endl; }
            try {
              caps[k] = read_capture(path, sw->opt.freq_start + 100e3 * (first + k));
              sw->fc_matches[first + k] = caps[k].fc_matches;
            } catch (const std::exception &e) {
              std::lock_guard<std::mutex> lk(sw->m);
              if (read_error.empty()) read_error = e.what();
            }
          }
        };
        const int n_readers = std::max(1, std::min(std::min(n, 8), (int)std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (int r = 1; r < n_readers; ++r) pool.push_back(std::thread(reader));
        reader();
        for (std::thread &t : pool) t.join();
        if (!read_error.empty()) throw std::runtime_error(read_error);
      }
      // raw-byte captures of one length go through the int8 path together; the rest one by one
      size_t n_cap = 0;
      for (int k = 0; k < n && !n_cap; ++k) if (!caps[k].iq_u8.empty()) n_cap = caps[k].samples.size();
      for (int k = 0; k < n; ++k) {
        const bool bytes = n_cap && !caps[k].iq_u8.empty() && caps[k].samples.size() == n_cap;
        sw->batched[first + k] = bytes;
        if (bytes) f.carriers.push_back(first + k);
        else { f.single_carriers.push_back(first + k); f.singles.push_back(Capture()); f.singles.back().samples.swap(caps[k].samples); }
      }
      if (!f.carriers.empty()) {
        const size_t need = f.carriers.size() * 2 * n_cap;
        if (need > pinned_bytes[slot]) {
          if (pinned[slot]) ctx[slot]->host_free(pinned[slot]);
          pinned[slot] = (unsigned char *)ctx[slot]->host_alloc(need);
          pinned_bytes[slot] = need;
        }
        std::vector<double> fcs(f.carriers.size());
        for (size_t j = 0; j < f.carriers.size(); ++j) {
          std::memcpy(pinned[slot] + j * 2 * n_cap, caps[f.carriers[j] - first].iq_u8.data(), 2 * n_cap);
          fcs[j] = sw->opt.freq_start + 100e3 * f.carriers[j];
        }
        ctx[slot]->enqueue_batch_host(pinned[slot], LCS_FMT_IQ_U8, (int)f.carriers.size(), (uint32_t)n_cap, sw->f_search_set, fcs, fcs,
                                     sw->fs_programmed);
      }
    }
    collect(slot);
    collect(slot ^ 1);
    for (int k = 0; k < 2; ++k) if (pinned[k]) ctx[k]->host_free(pinned[k]);
  } catch (const std::exception &e) {
    sw->fail(e.what());
  }
}

}  // namespace

int main(int argc, char *const argv[]) {
  Sweep sw;
  sw.opt = scan_args(argc, argv);
  Options &opt = sw.opt;
  validate(opt);
  if (!opt.load) {
    std::cerr << "Error: this build has no RTL-SDR support (GPU node); use --load with capbuf_XXXX.it files" << std::endl;
    return 1;
  }
  sw.fs_programmed = 1.92e6 * opt.correction;   // recorded-data convention, src/LTE-Tracker.cpp:791

  // frequency-offset hypotheses and carrier raster (src/CellSearch.cpp:463-465; n_extra uses freq_start only)
  const int n_extra = (int)std::floor((opt.freq_start * opt.ppm / 1e6 + 2.5e3) / 5e3);
  sw.f_search_set.set_size(2 * n_extra + 1);
  for (int i = 0; i <= 2 * n_extra; ++i) sw.f_search_set(i) = 5000.0 * (i - n_extra);
  sw.n_fc = (int)std::floor((opt.freq_end - opt.freq_start) / 100e3) + 1;
  sw.kBatch = (int)opt.batch;
  sw.n_batches = (sw.n_fc + sw.kBatch - 1) / sw.kBatch;
  sw.detected.resize(sw.n_fc);
  sw.fc_matches.assign(sw.n_fc, 1);
  sw.batched.assign(sw.n_fc, 0);
  sw.batch_done.assign(sw.n_batches, 0);

  std::vector<int> devices;
  if (opt.gpu_all) {
    const int n = lcs::Searcher::device_count();
    if (n < 1) { std::cerr << "Error: lcs_create failed (an MI355X is required; there is no CPU fallback)" << std::endl; return 2; }
    for (int d = 0; d < std::min(n, sw.n_batches); ++d) devices.push_back(d);
  } else if (!opt.gpu_list.empty()) {
    // one device thread (two contexts, two batches in flight) per list entry; a repeated index puts several threads on one
    // GPU -- the merge path of -g all on a one-GPU box
    const int n = lcs::Searcher::device_count();
    for (long d : opt.gpu_list) {
      if (d >= n) { std::cerr << "Error: gpu index " << d << " but only " << n << " GPU(s) are visible" << std::endl; return 2; }
      devices.push_back((int)d);
    }
  } else {
    devices.push_back((int)opt.gpu);
  }
  std::vector<std::thread> workers;
  for (size_t d = 0; d < devices.size(); ++d) workers.push_back(std::thread(device_thread, &sw, devices[d], (int)d, (int)devices.size()));

  // the reference's per-carrier report, in carrier order, as the batches complete
  bool failed = false;
  for (int b = 0; b < sw.n_batches && !failed; ++b) {
    {
      std::unique_lock<std::mutex> lk(sw.m);
      sw.cv.wait(lk, [&] { return sw.batch_done[b] || !sw.failure.empty(); });
      failed = !sw.failure.empty();
    }
    if (failed) break;
    std::lock_guard<std::mutex> lk(sw.m);                     // keeps the -v file messages of the workers out of a report
    for (int fci = b * sw.kBatch; fci < std::min(sw.n_fc, (b + 1) * sw.kBatch); ++fci) {
      const double fc_requested = opt.freq_start + 100e3 * fci;
      if (opt.verbosity >= 1) std::cout << "Examining center frequency " << num(fc_requested / 1e6) << " MHz ..." << std::endl;
      if (!sw.fc_matches[fci])
        std::cout << "Warning: while reading capture buffer " << fci << ", the read\n"
                  << "center frequency did not match the expected center frequency." << std::endl;
      if (opt.verbosity >= 2) std::cout << "  PSS correlation, peak search, SSS, FOE, TFG and MIB decoding ran on the GPU ("
                                        << (sw.batched[fci] ? "int8 batch" : "single buffer") << ")" << std::endl;
      if (opt.verbosity >= 1) announce(sw.detected[fci]);
    }
  }
  for (std::thread &t : workers) t.join();
  if (!sw.failure.empty()) {
    std::cerr << "Error: " << sw.failure << std::endl;
    return 2;
  }

  const std::vector<Cell> cells = merge_duplicates(sw.detected);
  if (cells.empty()) {
    std::cout << "No LTE cells were found..." << std::endl;
  } else {
    std::cout << "Detected the following cells:\n"
              << "A: #antenna ports C: CP type ; P: PHICH duration ; PR: PHICH resource type\n"
              << "CID A      fc   foff RXPWR C nRB P  PR CrystalCorrectionFactor\n";
    for (const Cell &c : cells) std::cout << table_row(c, opt.correction) << "\n";
    std::cout.flush();
  }
  return 0;
}
