// itfile.hpp -- minimal reader/writer for IT++ it_file (version 3) containers, C++ side.
// Format derived by inspection of the reference's data files; the CLI
// needs exactly what src/capbuf.cpp:104-114 reads and :191-196 writes: `capbuf` (dcvec) and
// `fc` (ivec).
#ifndef LCS_HOST_ITFILE_HPP
#define LCS_HOST_ITFILE_HPP

#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace itfile {

struct Var {
  std::string type;
  std::vector<char> data;
};

inline std::map<std::string, Var> read_all(const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("cannot open " + path);
  std::vector<char> buf;
  char tmp[1 << 16];
  size_t n;
  while ((n = std::fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
  std::fclose(f);
  if (buf.size() < 5 || std::memcmp(buf.data(), "IT++\x03", 5) != 0) throw std::runtime_error(path + ": not an IT++ v3 file");
  std::map<std::string, Var> out;
  size_t pos = 5;
  while (pos + 24 <= buf.size()) {
    uint64_t hdr, dat, blk;
    std::memcpy(&hdr, &buf[pos], 8);
    std::memcpy(&dat, &buf[pos + 8], 8);
    std::memcpy(&blk, &buf[pos + 16], 8);
    const char *p = &buf[pos + 24];
    std::string name(p);
    p += name.size() + 1;
    std::string type(p);
    if (pos + hdr + dat > buf.size()) throw std::runtime_error(path + ": truncated block");
    Var v;
    v.type = type;
    v.data.assign(buf.begin() + pos + hdr, buf.begin() + pos + hdr + dat);
    out[name] = v;
    if (blk == 0) break;
    pos += blk;
  }
  return out;
}

inline std::vector<std::complex<double> > get_dcvec(const std::map<std::string, Var> &m, const std::string &name) {
  std::map<std::string, Var>::const_iterator it = m.find(name);
  if (it == m.end() || it->second.type != "dcvec") throw std::runtime_error("variable " + name + " (dcvec) not found");
  uint64_t n;
  std::memcpy(&n, it->second.data.data(), 8);
  std::vector<std::complex<double> > v(n);
  std::memcpy(v.data(), it->second.data.data() + 8, n * 16);
  return v;
}

inline std::vector<int32_t> get_ivec(const std::map<std::string, Var> &m, const std::string &name) {
  std::map<std::string, Var>::const_iterator it = m.find(name);
  if (it == m.end() || it->second.type != "ivec") throw std::runtime_error("variable " + name + " (ivec) not found");
  uint64_t n;
  std::memcpy(&n, it->second.data.data(), 8);
  std::vector<int32_t> v(n);
  std::memcpy(v.data(), it->second.data.data() + 8, n * 4);
  return v;
}

inline void put_block(FILE *f, const std::string &name, const std::string &type, const std::vector<char> &payload) {
  const uint64_t hdr = 24 + name.size() + 1 + type.size() + 1 + 1, dat = payload.size(), blk = hdr + dat;
  std::fwrite(&hdr, 8, 1, f); std::fwrite(&dat, 8, 1, f); std::fwrite(&blk, 8, 1, f);
  std::fwrite(name.c_str(), 1, name.size() + 1, f);
  std::fwrite(type.c_str(), 1, type.size() + 1, f);
  std::fputc(0, f);
  std::fwrite(payload.data(), 1, payload.size(), f);
}

inline void write_capbuf(const std::string &path, const std::vector<std::complex<double> > &capbuf, int32_t fc) {
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) throw std::runtime_error("cannot create " + path);
  std::fwrite("IT++\x03", 1, 5, f);
  std::vector<char> p(8 + capbuf.size() * 16);
  const uint64_t n = capbuf.size();
  std::memcpy(p.data(), &n, 8);
  std::memcpy(p.data() + 8, capbuf.data(), capbuf.size() * 16);
  put_block(f, "capbuf", "dcvec", p);
  std::vector<char> q(8 + 4);
  const uint64_t one = 1;
  std::memcpy(q.data(), &one, 8);
  std::memcpy(q.data() + 8, &fc, 4);
  put_block(f, "fc", "ivec", q);
  std::fclose(f);
}

}  // namespace itfile
#endif
