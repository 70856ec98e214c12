// A C++ caller of include/scema_cluster.h, standing in for FEProblem::spline_building / spline_comparison
// (FE_problem.h:1196-1270): the host-side pieces (spline fit, similarity lists from a given distance matrix, greedy
// cover) need no GPU, so this runs anywhere; the distance kernel itself is exercised by tests/test_cluster.py -m gpu.
//
//   usage: scema_cluster_harness          prints "mapping i -> j" lines for a small hand-made case
#include <cmath>
#include <cstdio>
#include <vector>

#include "scema_cluster.h"

int main() {
  const int n = 4, steps = 5, npts = 6, num_gps = 6;
  const int32_t ids[n] = {0, 2, 3, 5};
  // histories 0 and 2 follow the same ramp, 3 a steeper one, 5 its own
  std::vector<double> hist((size_t)n * steps * 6, 0.0);
  const double slope[n] = {1.0e-3, 1.01e-3, 4.0e-3, -2.0e-3};
  for (int h = 0; h < n; h++)
    for (int s = 0; s < steps; s++) hist[((size_t)h * steps + s) * 6 + 2] = slope[h] * s;   // zz component
  std::vector<double> spline((size_t)n * npts * 6);
  if (scema_hist_splinify(hist.data(), n, steps, npts, spline.data())) return 1;
  // a straight line through equidistant knots is reproduced exactly by the natural spline
  for (int p = 0; p < npts; p++) {
    const double expect = slope[0] * (steps - 1) * p / (npts - 1);
    if (std::fabs(spline[(size_t)p * 6 + 2] - expect) > 1e-15) { fprintf(stderr, "spline value %d off\n", p); return 2; }
  }
  // distances on the host here (the product computes them on the GPU: scema_hist_compare)
  std::vector<double> diff((size_t)n * n, 0.0);
  for (int a = 0; a < n; a++)
    for (int b = 0; b < n; b++) {
      double sum = 0.0;
      for (int k = 0; k < npts * 6; k++) { const double d = spline[(size_t)a * npts * 6 + k] - spline[(size_t)b * npts * 6 + k]; sum += d * d; }
      diff[(size_t)a * n + b] = std::sqrt(sum);
    }
  std::vector<int64_t> start(n + 1);
  std::vector<int32_t> other(n * n);
  std::vector<double> dist(n * n);
  if (scema_hist_similar(diff.data(), n, 1.0e-3, n * n, start.data(), other.data(), dist.data())) return 3;
  std::vector<int32_t> edges;
  for (int a = 0; a < n; a++)
    for (int64_t e = start[a]; e < start[a + 1]; e++) { edges.push_back(ids[a]); edges.push_back(ids[other[e]]); }
  std::vector<int32_t> mapping(num_gps);
  if (scema_hist_cover(edges.data(), (int64_t)edges.size() / 2, num_gps, mapping.data())) return 4;
  for (int i = 0; i < num_gps; i++) printf("mapping %d -> %d\n", i, mapping[i]);
  return 0;
}
