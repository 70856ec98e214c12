// Test infrastructure only.  A driver around the REFERENCE's own cubic-spline header (headers/spline.h, the tk::spline
// class that MatHistPredict::Strain6D::splinify uses, strain2spline.h:140-180), compiled from where it lies under
// /root/reference by ref_build.mk into oracle/_ref/libspline_ref.so.  It exists to pin oracle/cluster_oracle.py and
// scema_amd/csrc/host/hist_cluster.cpp: no reference source is copied into this repository.
#include <vector>

#include "spline.h"   // -I/root/reference/headers

extern "C" {

// one strain component sampled at `steps` equidistant times in [0,1] -> npts equidistant spline values (the loop of
// strain2spline.h:153-179 for one component)
int ref_splinify_component(const double *y, int steps, int npts, double *out) {
  if (steps < 3 || npts < 2) return 1;
  std::vector<double> T(steps), Y(y, y + steps);
  for (int n = 0; n < steps; n++) T[n] = (double)n / (double)(steps - 1);
  tk::spline s;
  s.set_points(T, Y);
  for (int n = 0; n < npts; n++) out[n] = s((double)n / (double)(npts - 1));
  return 0;
}

}  // extern "C"
