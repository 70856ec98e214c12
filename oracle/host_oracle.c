/* host_oracle.c -- see host_oracle.h.  TEST INFRASTRUCTURE ONLY. */
#include "host_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static const int RAW_OF[3][3] = {{0, 3, 4}, {3, 1, 5}, {4, 5, 2}};  /* deal.II raw entry of (k,l) */
static const int FILE_OF[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}}; /* file line of (k,l), k<=l */

void ho_rotation_tensor(const double a[3], const double b[3], double R[9]) {
  double K[3][3], K2[3][3];
  double ccos = a[0] * b[0] + a[1] * b[1] + a[2] * b[2];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) K[i][j] = a[j] * b[i] - a[i] * b[j];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      K2[i][j] = 0;
      for (int k = 0; k < 3; k++) K2[i][j] += K[i][k] * K[k][j];
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[3 * i + j] = (i == j ? 1.0 : 0.0) + K[i][j] + (1 / (1 + ccos)) * K2[i][j];
}

void ho_rotate_sym2(const double t[6], const double R[9], double out[6]) {
  double T[3][3], tmp[3][3], tmp2[3][3];
  for (int k = 0; k < 3; k++)
    for (int l = 0; l < 3; l++) T[k][l] = t[RAW_OF[k][l]];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      tmp[i][j] = 0;
      for (int k = 0; k < 3; k++) tmp[i][j] += R[3 * i + k] * T[k][j];
    }
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      tmp2[i][j] = 0;
      for (int k = 0; k < 3; k++) tmp2[i][j] += tmp[i][k] * R[3 * j + k];
    }
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) out[RAW_OF[k][l]] = 0.5 * (tmp2[k][l] + tmp2[l][k]);
}

void ho_rotate_sym4(const double c[36], const double R[9], double out[36]) {
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++)
      for (int s = 0; s < 3; s++)
        for (int t = s; t < 3; t++) {
          double acc = 0;
          for (int m = 0; m < 3; m++)
            for (int n = 0; n < 3; n++)
              for (int p = 0; p < 3; p++)
                for (int r = 0; r < 3; r++)
                  acc += c[FILE_OF[m][n] * 6 + FILE_OF[p][r]] * R[3 * k + m] * R[3 * l + n] * R[3 * s + p] * R[3 * t + r];
          out[FILE_OF[k][l] * 6 + FILE_OF[s][t]] = acc;
        }
}

void ho_prepare_strain(const double eps[6], const double rotam[9], const double L0[3], int hooke, double out[6]) {
  double Rt[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) Rt[3 * i + j] = rotam[3 * j + i];
  ho_rotate_sym2(eps, Rt, out);
  if (!hooke)
    for (int j = 0; j < 3; j++) {
      out[RAW_OF[j][j]] *= L0[j];
      out[RAW_OF[j][(j + 1) % 3]] *= L0[(j + 2) % 3];
    }
}

void ho_hooke(const double c[36], const double eps[6], double out[6]) {
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) {
      double acc = 0;
      for (int m = 0; m < 3; m++)
        for (int n = 0; n < 3; n++) acc += c[FILE_OF[k][l] * 6 + FILE_OF[m][n]] * eps[RAW_OF[m][n]];
      out[RAW_OF[k][l]] = acc;
    }
}

void ho_store(int nrepl, const double *stress, const double *init_stress, const double *rotam, int hooke, double out[6]) {
  double acc[6] = {0, 0, 0, 0, 0, 0};
  for (int r = 0; r < nrepl; r++) {
    double loc[6], rot[6];
    for (int k = 0; k < 6; k++) loc[k] = stress[6 * r + k] - (hooke ? 0.0 : init_stress[6 * r + k]);
    ho_rotate_sym2(loc, rotam + 9 * r, rot);
    for (int k = 0; k < 6; k++) acc[k] += rot[k];
  }
  for (int k = 0; k < 6; k++) out[k] = acc[k] / nrepl;
}

int ho_read_sym2(const char *path, double out[6]) {
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  char line[1024];
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++)
      if (fgets(line, sizeof line, fp)) out[RAW_OF[k][l]] = strtod(line, NULL);
  fclose(fp);
  return 0;
}
int ho_write_sym2(const char *path, const double raw[6]) {
  FILE *fp = fopen(path, "w");
  if (!fp) return -1;
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++) fprintf(fp, "%.16g\n", raw[RAW_OF[k][l]]);
  fclose(fp);
  return 0;
}
int ho_read_sym4(const char *path, double out[36]) {
  FILE *fp = fopen(path, "r");
  if (!fp) return -1;
  char line[1024];
  memset(out, 0, 36 * sizeof(double));
  for (int k = 0; k < 3; k++)
    for (int l = k; l < 3; l++)
      for (int m = 0; m < 3; m++)
        for (int n = m; n < 3; n++)
          if (fgets(line, sizeof line, fp)) out[FILE_OF[k][l] * 6 + FILE_OF[m][n]] = strtod(line, NULL);
  fclose(fp);
  return 0;
}
