// tensors.h -- the four deal.II tensor types the hot path touches, as plain structs.
// SymmetricTensor<2,3> raw entry order xx,yy,zz,xy,xz,yz (what access_raw_entry(i) walks,
// reference stmd_sync.h:917-920); SymmetricTensor<4,3> kept as 6x6 in init.*.stiff file order
// (kl, mn each 00,01,02,11,12,22; reference read_write.h:149-171).
#pragma once
#include <array>
#include <cmath>

namespace scema {

struct Tensor1 {
  double v[3] = {0, 0, 0};
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};

struct Tensor2 {
  double m[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  static Tensor2 identity() {
    Tensor2 t;
    for (int i = 0; i < 3; i++) t.m[i][i] = 1.0;
    return t;
  }
  Tensor2 transposed() const {
    Tensor2 t;
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) t.m[i][j] = m[j][i];
    return t;
  }
};

struct SymmetricTensor2 {
  double raw[6] = {0, 0, 0, 0, 0, 0};
  static int raw_index(int k, int l) {
    static const int idx[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}};
    return idx[k][l];
  }
  SymmetricTensor2() = default;
  explicit SymmetricTensor2(const double a[6]) {
    for (int i = 0; i < 6; i++) raw[i] = a[i];
  }
  double &operator()(int k, int l) { return raw[raw_index(k, l)]; }
  double operator()(int k, int l) const { return raw[raw_index(k, l)]; }
  double &access_raw_entry(int i) { return raw[i]; }
  double access_raw_entry(int i) const { return raw[i]; }
  double norm() const {  // Frobenius norm of the full 3x3 tensor
    return std::sqrt(raw[0] * raw[0] + raw[1] * raw[1] + raw[2] * raw[2] + 2.0 * (raw[3] * raw[3] + raw[4] * raw[4] + raw[5] * raw[5]));
  }
  SymmetricTensor2 &operator+=(const SymmetricTensor2 &o) {
    for (int i = 0; i < 6; i++) raw[i] += o.raw[i];
    return *this;
  }
  SymmetricTensor2 &operator-=(const SymmetricTensor2 &o) {
    for (int i = 0; i < 6; i++) raw[i] -= o.raw[i];
    return *this;
  }
  SymmetricTensor2 &operator/=(double d) {
    for (int i = 0; i < 6; i++) raw[i] /= d;
    return *this;
  }
};

struct SymmetricTensor4 {
  double c[36] = {0};
  static int file_index(int k, int l) {
    static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return idx[k][l];
  }
  double &operator()(int k, int l, int m, int n) { return c[file_index(k, l) * 6 + file_index(m, n)]; }
  double operator()(int k, int l, int m, int n) const { return c[file_index(k, l) * 6 + file_index(m, n)]; }
  SymmetricTensor4 &operator+=(const SymmetricTensor4 &o) {
    for (int i = 0; i < 36; i++) c[i] += o.c[i];
    return *this;
  }
  SymmetricTensor4 &operator/=(double d) {
    for (int i = 0; i < 36; i++) c[i] /= d;
    return *this;
  }
};

// sigma = C : eps  (reference stmd_problem.h:386-392)
inline SymmetricTensor2 contract(const SymmetricTensor4 &C, const SymmetricTensor2 &e) {
  SymmetricTensor2 s;
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) {
      double acc = 0.0;
      for (int m = 0; m < 3; m++)
        for (int n = 0; n < 3; n++) acc += C(k, l, m, n) * e(m, n);
      s(k, l) = acc;
    }
  return s;
}

}  // namespace scema
