// Accuracy of the FP64 elementary functions that the ReaxFF kernels use instead of the library's (reax/rx_core.h: rx_log, rx_pow, rx_rcp,
// rx_rsqrt, rx_sqrt) and of the packed matrix entry, compiled for the host with the device algorithms (RX_DEVICE_MATH_ON_HOST: single-precision seeds stand in for the
// hardware estimates v_rcp_f64 / v_rsq_f64, which are at least as accurate).  Prints the worst errors over n pseudo-random arguments
// against long double references; tests/test_reax_math.py holds them against the bounds DESIGN.md states.
#define RX_HOST_TEST
#define RX_DEVICE_MATH_ON_HOST
#include "../scema_amd/csrc/reax/rx_core.h"

#include <cstdio>
#include <cstdlib>

static double ulps(double got, long double ref) {
  const double r = (double)ref;
  const double u = nextafter(fabs(r), 1e300) - fabs(r);
  return (double)(fabsl((long double)got - ref) / (long double)u);
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 2000000;
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0); };
  double w_log = 0, w_rcp = 0, w_rsq = 0, w_sqrt = 0, w_pow = 0, w_icbrt = 0;
  for (long i = 0; i < n; i++) {
    const double u = rnd();
    // distances squared, bond orders, sums of exponentials: 1e-9 .. 1e9, with a third of the arguments around 1 (where log loses relative accuracy)
    const double x = (i % 3 == 0) ? 0.5 + u : exp((u - 0.5) * 41.0);
    const double e = ulps(rx_log(x), logl((long double)x));
    if (fabs(log(x)) > 1e-3 && e > w_log) w_log = e;   // (relative to the result; next to x = 1 the absolute error stays below 2e-16)
    if (fabs(log(x)) <= 1e-3 && fabs(rx_log(x) - (double)logl((long double)x)) > 4e-16) w_log = 1e9;
    w_rcp = fmax(w_rcp, ulps(rx_rcp(x), 1.0L / (long double)x));
    w_rsq = fmax(w_rsq, ulps(rx_rsqrt(x), 1.0L / sqrtl((long double)x)));
    w_sqrt = fmax(w_sqrt, ulps(rx_sqrt(x), sqrtl((long double)x)));
    const double b = 0.001 + 3.0 * u, p = 0.5 + 8.0 * rnd();   // bond orders to the exponents of the force field
    w_pow = fmax(w_pow, fabs((double)(((long double)rx_pow(b, p) - powl((long double)b, (long double)p)) / powl((long double)b, (long double)p))));
    (void)w_icbrt;
  }
  // the packed matrix entry (rx_hpack / rx_hunpack): the column comes back, the value within 2^-37 of itself (round to nearest of the upper 48 bits)
  double w_pack = 0;
  long pack_bad = 0;
  for (long i = 0; i < n / 4; i++) {
    const double h = exp((rnd() - 0.5) * 20.0);   // shielded Coulomb terms: 1e-4 .. 1e4 kcal/mol/e^2
    const int col = (int)(rnd() * 65536.0) & 0xFFFF;
    int c2;
    const double h2 = rx_hunpack(rx_hpack(h, col), &c2);
    if (c2 != col) pack_bad++;
    w_pack = fmax(w_pack, fabs(h2 - h) / h);
  }
  { int c2; if (rx_hunpack(rx_hpack(1.9999999999999998, 65535), &c2) != 2.0 || c2 != 65535) pack_bad++; }   // (a rounding that carries into the exponent)
  printf("{\"n\": %ld, \"pack_rel\": %.3e, \"pack_bad\": %ld, \"log_ulp\": %.3f, \"rcp_ulp\": %.3f, \"rsqrt_ulp\": %.3f, \"sqrt_ulp\": %.3f, \"pow_rel\": %.3e}\n", n, w_pack, pack_bad, w_log, w_rcp, w_rsq, w_sqrt, w_pow);
  return 0;
}
