// hist_cluster.cpp -- host side of the strain-history clustering step (include/scema_cluster.h):
// spline fit (strain2spline.h:140-180 on tk::spline, spline.h:293-394), similarity lists (strain2spline.h:265-314,
// 601-612) and the greedy cover of clustering/coarsegrain_dependency_network.py:24-95.  The all-pairs distances are the
// HIP kernel in md_cluster.hip.  Compiled without floating-point contraction: the spline arithmetic is kept in the
// reference's order so that the values agree bit for bit (tests/golden/cluster_golden.json).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/scema_cluster.h"
#include "../../../include/scema_md.h"

namespace {

// natural cubic spline through (i/(steps-1), y[i]); values at n/(npts-1)
void spline_component(const double *y, int stride, int steps, int npts, double *out, int ostride, std::vector<double> &w) {
  const int n = steps;
  w.assign((size_t)9 * n, 0.0);
  double *x = &w[0], *lo = &w[n], *di = &w[2 * n], *up = &w[3 * n], *rhs = &w[4 * n], *inv = &w[5 * n], *b = &w[6 * n], *a = &w[7 * n],
         *c = &w[8 * n];
  for (int i = 0; i < n; i++) x[i] = (double)i / (double)(n - 1);
  auto Y = [&](int i) { return y[(size_t)i * stride]; };
  for (int i = 1; i < n - 1; i++) {
    lo[i] = 1.0 / 3.0 * (x[i] - x[i - 1]);
    di[i] = 2.0 / 3.0 * (x[i + 1] - x[i - 1]);
    up[i] = 1.0 / 3.0 * (x[i + 1] - x[i]);
    rhs[i] = (Y(i + 1) - Y(i)) / (x[i + 1] - x[i]) - (Y(i) - Y(i - 1)) / (x[i] - x[i - 1]);
  }
  di[0] = 2.0; up[0] = 0.0; rhs[0] = 0.0;   // zero curvature at both ends (tk::spline default)
  di[n - 1] = 2.0; lo[n - 1] = 0.0; rhs[n - 1] = 0.0;
  for (int i = 0; i < n; i++) {             // rows scaled by the inverse diagonal
    inv[i] = 1.0 / di[i];
    lo[i] *= inv[i]; up[i] *= inv[i]; di[i] = 1.0;
  }
  for (int k = 0; k < n - 1; k++) {         // elimination of the sub-diagonal
    const double f = -lo[k + 1] / di[k];
    lo[k + 1] = -f;
    di[k + 1] = di[k + 1] + f * up[k];
  }
  // forward substitution into c (scratch), back substitution into b
  for (int i = 0; i < n; i++) {
    double s = 0.0;
    if (i > 0) s += lo[i] * c[i - 1];
    c[i] = rhs[i] * inv[i] - s;
  }
  for (int i = n - 1; i >= 0; i--) {
    double s = 0.0;
    if (i < n - 1) s += up[i] * b[i + 1];
    b[i] = (c[i] - s) / di[i];
  }
  for (int i = 0; i < n - 1; i++) {
    a[i] = 1.0 / 3.0 * (b[i + 1] - b[i]) / (x[i + 1] - x[i]);
    c[i] = (Y(i + 1) - Y(i)) / (x[i + 1] - x[i]) - 1.0 / 3.0 * (2.0 * b[i] + b[i + 1]) * (x[i + 1] - x[i]);
  }
  const double hl = x[n - 1] - x[n - 2];
  a[n - 1] = 0.0;
  c[n - 1] = 3.0 * a[n - 2] * hl * hl + 2.0 * b[n - 2] * hl + c[n - 2];
  for (int p = 0; p < npts; p++) {
    const double t = (double)p / (double)(npts - 1);
    int idx = (int)(std::lower_bound(x, x + n, t) - x) - 1;
    if (idx < 0) idx = 0;
    const double h = t - x[idx];
    double v;
    if (t < x[0]) v = (b[0] * h + c[0]) * h + Y(0);
    else if (t > x[n - 1]) v = (b[n - 1] * h + c[n - 1]) * h + Y(n - 1);
    else v = ((a[idx] * h + b[idx]) * h + c[idx]) * h + Y(idx);
    out[(size_t)p * ostride] = v;
  }
}

}  // namespace

extern "C" {

int scema_hist_splinify(const double *hist, int32_t n, int32_t steps, int32_t npts, double *spline) {
  if (!hist || !spline || n < 0 || steps < 3 || npts < 2) return SCEMA_MD_ERR_ARG;
  std::vector<double> w;
  for (int32_t h = 0; h < n; h++)
    for (int k = 0; k < 6; k++)
      spline_component(hist + (size_t)h * steps * 6 + k, 6, steps, npts, spline + (size_t)h * npts * 6 + k, 6, w);
  return SCEMA_MD_OK;
}

int scema_hist_similar(const double *diff, int32_t n, double threshold, int64_t capacity, int64_t *start, int32_t *other, double *dist) {
  if (!diff || !start || n < 0) return SCEMA_MD_ERR_ARG;
  // history a collects (b, diff) from the pairs (b < a) first -- pushed while b's row was walked -- then from (a, b > a):
  // ascending b either way
  int64_t m = 0;
  for (int32_t a = 0; a < n; a++) {
    start[a] = m;
    for (int32_t b = 0; b < n; b++) {
      if (b == a) continue;
      const double v = (a < b) ? diff[(size_t)a * n + b] : diff[(size_t)b * n + a];
      if (v < threshold) {
        if (m < capacity && other && dist) { other[m] = b; dist[m] = v; }
        m++;
      }
    }
  }
  start[n] = m;
  return m <= capacity ? SCEMA_MD_OK : SCEMA_MD_ERR_ARG;
}

int scema_hist_cover(const int32_t *edges, int64_t m, int32_t num_gps, int32_t *mapping) {
  if (!mapping || num_gps < 0 || m < 0 || (m > 0 && !edges)) return SCEMA_MD_ERR_ARG;
  for (int32_t i = 0; i < num_gps; i++) mapping[i] = i;
  // nodes in the order they enter the graph; adjacency as sorted vectors
  std::unordered_map<int32_t, int32_t> index;
  std::vector<int32_t> node;
  std::vector<std::vector<int32_t>> adj;
  auto idx = [&](int32_t c) {
    auto it = index.find(c);
    if (it != index.end()) return it->second;
    const int32_t k = (int32_t)node.size();
    index.emplace(c, k);
    node.push_back(c);
    adj.emplace_back();
    return k;
  };
  for (int64_t e = 0; e < m; e++) {
    const int32_t c1 = edges[2 * e], c2 = edges[2 * e + 1];
    if (c1 < 0 || c1 >= num_gps || c2 < 0 || c2 >= num_gps) return SCEMA_MD_ERR_ARG;
    const int32_t a = idx(c1), b = idx(c2);
    if (a == b) continue;   // the reference never writes a history's own id into its file
    adj[a].push_back(b);
    adj[b].push_back(a);
  }
  const int32_t nn = (int32_t)node.size();
  std::vector<int32_t> deg(nn, 0);
  for (int32_t k = 0; k < nn; k++) {
    std::sort(adj[k].begin(), adj[k].end());
    adj[k].erase(std::unique(adj[k].begin(), adj[k].end()), adj[k].end());   // every edge is listed by both of its files
    deg[k] = (int32_t)adj[k].size();
  }
  std::vector<char> alive(nn, 1);
  int32_t remaining = nn;
  while (remaining > 0) {
    // highest degree; among equals the node that entered the graph last (stable ascending sort, last element)
    int32_t best = -1, bestdeg = -1;
    for (int32_t k = 0; k < nn; k++)
      if (alive[k] && deg[k] >= bestdeg) { best = k; bestdeg = deg[k]; }
    mapping[node[best]] = node[best];
    alive[best] = 0;
    remaining--;
    std::vector<int32_t> gone(1, best);
    for (int32_t nb : adj[best])
      if (alive[nb]) { mapping[node[nb]] = node[best]; alive[nb] = 0; remaining--; gone.push_back(nb); }
    for (int32_t g : gone)
      for (int32_t nb : adj[g])
        if (alive[nb]) deg[nb]--;
  }
  return SCEMA_MD_OK;
}

int scema_hist_cluster(const int32_t *ids, const double *hist, int32_t n, int32_t steps, int32_t npts, double threshold, int32_t num_gps,
                       int32_t device, int32_t *mapping) {
  if (!mapping || n < 0 || (n > 0 && (!ids || !hist))) return SCEMA_MD_ERR_ARG;
  if (n == 0) return scema_hist_cover(nullptr, 0, num_gps, mapping);
  std::vector<double> spline((size_t)n * npts * 6);
  int rc = scema_hist_splinify(hist, n, steps, npts, spline.data());
  if (rc) return rc;
  // similar pairs straight from the GPU (the n x n matrix is never formed); grow the list once if it was too small
  int64_t cap = std::max<int64_t>(1024, (int64_t)n * 16), cnt = 0;
  std::vector<int32_t> pr;
  std::vector<double> ds;
  for (int attempt = 0; attempt < 2; attempt++) {
    pr.assign((size_t)2 * cap, 0);
    ds.assign((size_t)cap, 0.0);
    rc = scema_hist_edges(spline.data(), n, npts * 6, threshold, device, cap, pr.data(), ds.data(), &cnt);
    if (rc != SCEMA_MD_ERR_OVERFLOW) break;
    cap = cnt;
  }
  if (rc) return rc;
  // the script reads file after file (ascending id here), each file listing the other histories in ascending position
  std::vector<std::pair<int32_t, int32_t>> dir;
  dir.reserve((size_t)2 * cnt);
  for (int64_t k = 0; k < cnt; k++) { dir.emplace_back(pr[2 * k], pr[2 * k + 1]); dir.emplace_back(pr[2 * k + 1], pr[2 * k]); }
  std::sort(dir.begin(), dir.end());
  const int64_t m = (int64_t)dir.size();
  std::vector<int32_t> edges((size_t)2 * std::max<int64_t>(m, 1));
  for (int64_t e = 0; e < m; e++) { edges[2 * e] = ids[dir[e].first]; edges[2 * e + 1] = ids[dir[e].second]; }
  return scema_hist_cover(edges.data(), m, num_gps, mapping);
}

}  // extern "C"
