// md_cluster.hip -- all-pairs L2 distances between splined strain histories (include/scema_cluster.h;
// compare_L2_norm + the pair loops of compare_histories_with_all_ranks, headers/strain2spline.h:469-487, 546-614).
//
// n histories of d = 6 * spline points doubles.  One workgroup of 256 threads owns a 64 x 64 tile of pairs of the upper
// triangle; the two 64-row panels are staged through the LDS in chunks of 32 columns (33 KB), every thread carries
// 4 x 4 pairs (8 LDS reads per 16 pair elements), and the sum over the d columns runs in ascending order with separate multiply and add (no contraction),
// which is the reference's arithmetic: results are bit-identical to the CPU loop.  The kernel reads n*d*8 bytes and writes
// n*n*8: for the sizes of the path (n = 576 ... 4 864, d = 60 ... 600) it is bound by the n^2 result write.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "../../include/scema_cluster.h"
#include "../../include/scema_md.h"

#define CT 64      // pair tile edge
#define CR (CT / 16) // pairs per thread and direction
#define CK 32      // columns staged per step

// EMIT = false: the full symmetric matrix.  EMIT = true: only the pairs a < b below the threshold, appended to a list
// (what the similarity files hold; the n^2 matrix never exists).
template <bool EMIT>
__global__ __launch_bounds__(256) void k_hist_compare(const double *__restrict__ sp, int n, int d, double *__restrict__ out, double threshold,
                                                      unsigned long long *__restrict__ count, unsigned long long cap, int2 *__restrict__ pairs) {
#pragma clang fp contract(off)   // multiply and add stay separate roundings, as in the reference's loop
  // upper-triangle tile (ti <= tj) from the linear block index
  int tj = (int)((sqrt(8.0 * (double)blockIdx.x + 1.0) - 1.0) * 0.5);
  while ((long long)(tj + 1) * (tj + 2) / 2 <= (long long)blockIdx.x) tj++;
  while ((long long)tj * (tj + 1) / 2 > (long long)blockIdx.x) tj--;
  const int ti = (int)((long long)blockIdx.x - (long long)tj * (tj + 1) / 2);
  __shared__ double s_raw[2 * CT * (CK + 1)];   // two panels; reused as a CT x (CT+1) transpose buffer for the mirror tile
  double (*s_a)[CK + 1] = (double (*)[CK + 1])s_raw;
  double (*s_b)[CK + 1] = (double (*)[CK + 1])(s_raw + CT * (CK + 1));
  static_assert(2 * CT * (CK + 1) >= CT * (CT + 1), "transpose buffer");
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;   // 16 x 16 threads, CR x CR pairs each
  const int a0 = ti * CT, b0 = tj * CT;
  double acc[CR][CR];
#pragma unroll
  for (int u = 0; u < CR; u++)
#pragma unroll
    for (int v = 0; v < CR; v++) acc[u][v] = 0.0;
  for (int k0 = 0; k0 < d; k0 += CK) {
    for (int e = threadIdx.x; e < CT * CK; e += 256) {
      const int r = e / CK, c = e % CK;
      s_a[r][c] = (a0 + r < n && k0 + c < d) ? sp[(size_t)(a0 + r) * d + k0 + c] : 0.0;
      s_b[r][c] = (b0 + r < n && k0 + c < d) ? sp[(size_t)(b0 + r) * d + k0 + c] : 0.0;
    }
    __syncthreads();
    const int kc = min(CK, d - k0);
    for (int c = 0; c < kc; c++) {
      double ra[CR], rb[CR];
#pragma unroll
      for (int u = 0; u < CR; u++) { ra[u] = s_a[ty + 16 * u][c]; rb[u] = s_b[tx + 16 * u][c]; }
#pragma unroll
      for (int u = 0; u < CR; u++)
#pragma unroll
        for (int v = 0; v < CR; v++) {
          const double df = ra[u] - rb[v];
          const double sq = df * df;
          acc[u][v] = acc[u][v] + sq;
        }
    }
    __syncthreads();
  }
  double (*s_t)[CT + 1] = (double (*)[CT + 1])s_raw;
#pragma unroll
  for (int u = 0; u < CR; u++)
#pragma unroll
    for (int v = 0; v < CR; v++) {
      const int la = ty + 16 * u, lb = tx + 16 * v;
      const int a = a0 + la, b = b0 + lb;
      const double r = __dsqrt_rn(acc[u][v]);   // correctly rounded
      if (!EMIT) {
        if (a < n && b < n) out[(size_t)a * n + b] = r;   // rows of the tile: consecutive b across tx
        s_t[lb][la] = r;                                  // the mirror tile goes through the LDS so that it is written row-wise too
      } else if (a < n && b < n && a < b && r < threshold) {
        const unsigned long long k = atomicAdd(count, 1ull);
        if (k < cap) { pairs[k] = make_int2(a, b); out[k] = r; }
      }
    }
  if (!EMIT && ti != tj) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CR; u++)
#pragma unroll
      for (int v = 0; v < CR; v++) {
        const int lb = ty + 16 * u, la = tx + 16 * v;
        const int a = a0 + la, b = b0 + lb;
        if (a < n && b < n) out[(size_t)b * n + a] = s_t[lb][la];
      }
  }
}

extern "C" {

int scema_hist_compare_device(const double *spline_dev, int32_t n, int32_t d, double *diff_dev, void *stream) {
  if (n < 0 || d <= 0 || (n > 0 && (!spline_dev || !diff_dev))) return SCEMA_MD_ERR_ARG;
  if (n == 0) return SCEMA_MD_OK;
  const long long nt = (n + CT - 1) / CT;
  const long long nblocks = nt * (nt + 1) / 2;
  if (nblocks > 0x7fffffffLL) return SCEMA_MD_ERR_ARG;
  hipLaunchKernelGGL(k_hist_compare<false>, dim3((unsigned)nblocks), dim3(256), 0, (hipStream_t)stream, spline_dev, n, d, diff_dev, 0.0, nullptr, 0ull, nullptr);
  return hipGetLastError() == hipSuccess ? SCEMA_MD_OK : SCEMA_MD_ERR_DEVICE;
}

int scema_hist_compare(const double *spline, int32_t n, int32_t d, double *diff, int32_t device) {
  if (n < 0 || d <= 0 || (n > 0 && (!spline || !diff))) return SCEMA_MD_ERR_ARG;
  if (n == 0) return SCEMA_MD_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return SCEMA_MD_ERR_DEVICE;   // no CPU fallback
  if (hipSetDevice(device) != hipSuccess) return SCEMA_MD_ERR_DEVICE;
  double *ds = nullptr, *dd = nullptr;
  int rc = SCEMA_MD_ERR_DEVICE;
  if (hipMalloc(&ds, (size_t)n * d * sizeof(double)) == hipSuccess && hipMalloc(&dd, (size_t)n * n * sizeof(double)) == hipSuccess &&
      hipMemcpy(ds, spline, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice) == hipSuccess) {
    rc = scema_hist_compare_device(ds, n, d, dd, nullptr);
    if (rc == SCEMA_MD_OK && (hipDeviceSynchronize() != hipSuccess ||
                              hipMemcpy(diff, dd, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess))
      rc = SCEMA_MD_ERR_DEVICE;
  }
  if (ds) (void)hipFree(ds);
  if (dd) (void)hipFree(dd);
  return rc;
}

int scema_hist_edges(const double *spline, int32_t n, int32_t d, double threshold, int32_t device, int64_t capacity, int32_t *pairs,
                     double *dist, int64_t *count) {
  if (n < 0 || d <= 0 || !count || capacity < 0 || (n > 0 && !spline) || (capacity > 0 && (!pairs || !dist))) return SCEMA_MD_ERR_ARG;
  *count = 0;
  if (n < 2) return SCEMA_MD_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return SCEMA_MD_ERR_DEVICE;   // no CPU fallback
  if (hipSetDevice(device) != hipSuccess) return SCEMA_MD_ERR_DEVICE;
  const long long nt = (n + CT - 1) / CT;
  const long long nblocks = nt * (nt + 1) / 2;
  if (nblocks > 0x7fffffffLL) return SCEMA_MD_ERR_ARG;
  double *ds = nullptr, *dd = nullptr;
  int2 *dp = nullptr;
  unsigned long long *dc = nullptr, hc = 0;
  const size_t cap = (size_t)std::max<int64_t>(capacity, 1);
  int rc = SCEMA_MD_ERR_DEVICE;
  if (hipMalloc(&ds, (size_t)n * d * sizeof(double)) == hipSuccess && hipMalloc(&dd, cap * sizeof(double)) == hipSuccess &&
      hipMalloc(&dp, cap * sizeof(int2)) == hipSuccess && hipMalloc(&dc, sizeof(unsigned long long)) == hipSuccess &&
      hipMemcpy(ds, spline, (size_t)n * d * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
      hipMemset(dc, 0, sizeof(unsigned long long)) == hipSuccess) {
    hipLaunchKernelGGL(k_hist_compare<true>, dim3((unsigned)nblocks), dim3(256), 0, nullptr, ds, n, d, dd, threshold, dc, (unsigned long long)capacity, dp);
    if (hipGetLastError() == hipSuccess && hipDeviceSynchronize() == hipSuccess &&
        hipMemcpy(&hc, dc, sizeof hc, hipMemcpyDeviceToHost) == hipSuccess) {
      *count = (int64_t)hc;
      const size_t m = (size_t)std::min<unsigned long long>(hc, (unsigned long long)capacity);
      rc = SCEMA_MD_OK;
      if (m > 0 && (hipMemcpy(pairs, dp, m * sizeof(int2), hipMemcpyDeviceToHost) != hipSuccess ||
                    hipMemcpy(dist, dd, m * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess))
        rc = SCEMA_MD_ERR_DEVICE;
      if (rc == SCEMA_MD_OK && hc > (unsigned long long)capacity) rc = SCEMA_MD_ERR_OVERFLOW;   // *count says how many there are
    }
  }
  if (ds) (void)hipFree(ds);
  if (dd) (void)hipFree(dd);
  if (dp) (void)hipFree(dp);
  if (dc) (void)hipFree(dc);
  return rc;
}

}  // extern "C"
