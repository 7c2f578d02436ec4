// lcs_containers.h -- the handful of IT++ container operations the searcher call surface needs.
//
// The reference passes itpp::vec / cvec / mat / imat / cmat by reference (include/searcher.h).
// IT++ is not available in this build environment, so the host code here uses these minimal
// stand-ins: same member names (length(), size(), rows(), cols(), operator()(i), operator()(r,c),
// set_size(), _data()), same memory layout (vectors contiguous, matrices COLUMN-major like
// itpp::Mat).  include/searcher_amd.h is templated on the container namespace, so building
// against real IT++ only needs `#define LCS_CONTAINER_NS itpp`.
#ifndef LCS_CONTAINERS_H
#define LCS_CONTAINERS_H

#include <complex>
#include <cstddef>
#include <vector>

namespace lcsc {

template <class T>
class Vec {
 public:
  Vec() {}
  explicit Vec(int n) : d_(n) {}
  int length() const { return (int)d_.size(); }
  int size() const { return (int)d_.size(); }
  void set_size(int n, bool copy = false) { (void)copy; d_.resize(n); }
  void set_length(int n, bool copy = false) { set_size(n, copy); }
  T &operator()(int i) { return d_[i]; }
  const T &operator()(int i) const { return d_[i]; }
  T &operator[](int i) { return d_[i]; }
  const T &operator[](int i) const { return d_[i]; }
  T *_data() { return d_.data(); }
  const T *_data() const { return d_.data(); }
  void del(int i) { d_.erase(d_.begin() + i); }       // itpp::Vec::del

 private:
  std::vector<T> d_;
};

template <class T>
class Mat {   // column-major, like itpp::Mat
 public:
  Mat() : r_(0), c_(0) {}
  Mat(int r, int c) : r_(r), c_(c), d_((size_t)r * c) {}
  int rows() const { return r_; }
  int cols() const { return c_; }
  void set_size(int r, int c, bool copy = false) { (void)copy; r_ = r; c_ = c; d_.resize((size_t)r * c); }
  T &operator()(int r, int c) { return d_[(size_t)c * r_ + r]; }
  const T &operator()(int r, int c) const { return d_[(size_t)c * r_ + r]; }
  T *_data() { return d_.data(); }
  const T *_data() const { return d_.data(); }

 private:
  int r_, c_;
  std::vector<T> d_;
};

typedef Vec<double> vec;
typedef Vec<int> ivec;
typedef Vec<std::complex<double> > cvec;
typedef Mat<double> mat;
typedef Mat<int> imat;
typedef Mat<std::complex<double> > cmat;

}  // namespace lcsc
#endif
