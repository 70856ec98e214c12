// math_calc.h -- rotation helpers of the hot path (reference headers/math_calc.h:23-99)
#pragma once
#include "tensors.h"

namespace scema {

// R = I + K + K^2 / (1 + a.b), K_ij = a_j b_i - a_i b_j   (reference math_calc.h:23-50)
inline Tensor2 compute_rotation_tensor(const Tensor1 &vorig, const Tensor1 &vdest) {
  double ccos = 0.0;
  for (int i = 0; i < 3; i++) ccos += vorig[i] * vdest[i];
  Tensor2 skew, sq, rot;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) skew.m[i][j] = vorig[j] * vdest[i] - vorig[i] * vdest[j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) sq.m[i][j] += skew.m[i][k] * skew.m[k][j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) rot.m[i][j] = (i == j ? 1.0 : 0.0) + skew.m[i][j] + (1 / (1 + ccos)) * sq.m[i][j];
  return rot;
}

// sym(R T R^T) with the explicit 0.5 (t_kl + t_lk)   (reference math_calc.h:52-71)
inline SymmetricTensor2 rotate_tensor(const SymmetricTensor2 &t, const Tensor2 &R) {
  double T[3][3], a[3][3] = {{0}}, b[3][3] = {{0}};
  for (int k = 0; k < 3; k++)
    for (int l = 0; l < 3; l++) T[k][l] = t(k, l);
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) a[i][j] += R.m[i][k] * T[k][j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) b[i][j] += a[i][k] * R.m[j][k];
  SymmetricTensor2 s;
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) s(k, l) = 0.5 * (b[k][l] + b[l][k]);
  return s;
}

// rank-4 rotation over the upper triangles   (reference math_calc.h:73-99)
inline SymmetricTensor4 rotate_tensor(const SymmetricTensor4 &C, const Tensor2 &R) {
  SymmetricTensor4 out;
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++)
      for (int s = 0; s < 3; s++)
        for (int t = s; t < 3; t++) {
          double acc = 0.0;
          for (int m = 0; m < 3; m++)
            for (int n = 0; n < 3; n++)
              for (int p = 0; p < 3; p++)
                for (int r = 0; r < 3; r++) acc += C(m, n, p, r) * R.m[k][m] * R.m[l][n] * R.m[s][p] * R.m[t][r];
          out(k, l, s, t) = acc;
        }
  return out;
}

}  // namespace scema
