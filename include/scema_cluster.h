/* scema_cluster.h -- C ABI of the strain-history clustering step: the part of
 * FEProblem::history_analysis (FE_problem.h:1196-1290) that decides which quadrature points need their own MD run.
 * Plain pointers and sizes; host buffers unless a name says "device".  Return 0 on success (SCEMA_MD_* codes of
 * scema_md.h otherwise).
 */
#ifndef SCEMA_CLUSTER_H
#define SCEMA_CLUSTER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Replaces MatHistPredict::Strain6D::splinify (headers/strain2spline.h:140-180) for a batch: natural cubic splines
 * (tk::spline defaults, headers/spline.h:293-394) through each of the 6 strain components of n histories of `steps`
 * entries (hist[n][steps][6], components xx,yy,zz,xy,xz,yz), sampled at npts equidistant parameters:
 * spline[n][npts*6], point-major.  steps >= 3 (strain2spline.h:145-148), npts >= 2. */
int scema_hist_splinify(const double *hist, int32_t n, int32_t steps, int32_t npts, double *spline);

/* Replaces the all-pairs comparison of compare_histories_with_all_ranks (strain2spline.h:546-614) with
 * compare_L2_norm (:469-487): diff[a*n+b] = sqrt(sum_k (spline_a[k] - spline_b[k])^2) for all a, b (symmetric, zero
 * diagonal), computed on HIP device `device` with the sum taken in the reference's order (bit-identical results).
 * spline[n][d] and diff[n][n] are host buffers; the _device form takes device pointers and a stream (void* hipStream_t). */
int scema_hist_compare(const double *spline, int32_t n, int32_t d, double *diff, int32_t device);
int scema_hist_compare_device(const double *spline_dev, int32_t n, int32_t d, double *diff_dev, void *stream);

/* The same comparison without the n x n matrix: only the pairs a < b with diff < threshold come back (pairs[2*k],
 * pairs[2*k+1] = indices into the history vector, dist[k]), in no particular order; *count = how many exist.  Returns
 * SCEMA_MD_ERR_OVERFLOW if capacity was too small (call again with capacity >= *count). */
int scema_hist_edges(const double *spline, int32_t n, int32_t d, double threshold, int32_t device, int64_t capacity, int32_t *pairs,
                     double *dist, int64_t *count);

/* Body of the files Strain6D::most_similar_histories_to_file writes (strain2spline.h:301-314): for history a, the
 * histories b with diff < threshold, in the order the single-rank loop pushes them (strain2spline.h:601-612).
 * CSR output: start[n+1], other[...] (index into the history vector), dist[...]; capacity = entries available in
 * other/dist; returns SCEMA_MD_ERR_ARG if it does not suffice (start[n] then holds the required count). */
int scema_hist_similar(const double *diff, int32_t n, double threshold, int64_t capacity, int64_t *start, int32_t *other,
                       double *dist);

/* Replaces clustering/coarsegrain_dependency_network.py:24-95: greedy cover of the similarity graph.  edges[2*m] are
 * the (cell1, cell2) ids in the order the script would read them (file by file, line by line); mapping[num_gps] gets
 * "where quadrature point i takes its MD result from" (itself if it is not in the graph), the content of mapping.csv. */
int scema_hist_cover(const int32_t *edges, int64_t m, int32_t num_gps, int32_t *mapping);

/* The whole step for the histories of the quadrature points to be updated (ids[n], ascending; the files are taken in
 * ascending id order, where the script takes them in the file system's glob order): splines, distances on the GPU,
 * similarity lists, cover.  mapping[num_gps] as above. */
int scema_hist_cluster(const int32_t *ids, const double *hist, int32_t n, int32_t steps, int32_t npts, double threshold,
                       int32_t num_gps, int32_t device, int32_t *mapping);

#ifdef __cplusplus
}
#endif
#endif
