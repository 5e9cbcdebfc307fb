// eval_harness.cpp — TEST INFRASTRUCTURE: the product's per-pair arithmetic (kube-batch_amd/csrc/kb_eval.hpp, compiled for the host
// through the HIP stand-in header) against the reference's own arithmetic written out with int64 `/` and IEEE double `/`
// (vendor/k8s.io/kubernetes/pkg/scheduler/algorithm/priorities/{least_requested,most_requested,balanced_resource_allocation}.go,
// api/resource_info.go:268-302).  tests/test_eval_core_cpu.py drives it: random, structured and near-boundary operands.
#include <math.h>
#include <stdint.h>

#include "../../kube-batch_amd/csrc/kb_eval.hpp"

namespace {
inline uint64_t mix(uint64_t &s) {   // splitmix64
  uint64_t z = (s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
// the Go code, literally
int64_t ref_least(int64_t req, int64_t cap) { return (cap == 0 || req > cap) ? 0 : ((cap - req) * 10) / cap; }
int64_t ref_most(int64_t req, int64_t cap) { return (cap == 0 || req > cap) ? 0 : (req * 10) / cap; }
double ref_frac(int64_t req, int64_t cap) { return cap == 0 ? 1.0 : (double)req / (double)cap; }
uint32_t ref_score(int64_t t_nzc, int64_t t_nzm, int64_t n_nzc, int64_t n_nzm, int64_t ac, int64_t am, int wL, int wM, int wB) {
  const int64_t rc = n_nzc + t_nzc, rm = n_nzm + t_nzm;
  const int least = (int)((ref_least(rc, ac) + ref_least(rm, am)) / 2), most = (int)((ref_most(rc, ac) + ref_most(rm, am)) / 2);
  const double cf = ref_frac(rc, ac), mf = ref_frac(rm, am);
  int bal = 0;
  if (!(cf >= 1.0 || mf >= 1.0)) bal = (int)(int64_t)((1.0 - fabs(cf - mf)) * 10.0);
  return (uint32_t)(least * wL + most * wM + bal * wB);
}
bool ref_le(double l, double r, double eps) { return l < r || fabs(l - r) < eps; }
}   // namespace

extern "C" {
double eh_div_small(double a, double b) { return div_small_f64(a, b, 1.0 / b); }
uint32_t eh_score(int64_t t_nzc, int64_t t_nzm, int64_t n_nzc, int64_t n_nzm, int64_t ac, int64_t am, int wL, int wM, int wB) {
  return score_core_f64((double)t_nzc, (double)t_nzm, (double)n_nzc, (double)n_nzm, (double)ac, (double)am, ac ? 1.0 / (double)ac : 0.0, am ? 1.0 / (double)am : 0.0, wL, wM, wB);
}
uint32_t eh_ref_score(int64_t t_nzc, int64_t t_nzm, int64_t n_nzc, int64_t n_nzm, int64_t ac, int64_t am, int wL, int wM, int wB) {
  return ref_score(t_nzc, t_nzm, n_nzc, n_nzm, ac, am, wL, wM, wB);
}
// mode 0: a, b uniform below 2^bits; 1: b with few significant bits / all ones / powers of two +-1, a = k b / 2^j +- small (quotients next to
// short binary fractions: the rounding boundaries' neighbourhood); 2: small denominators (every quotient a short repeating fraction)
uint64_t eh_div_mismatches(uint64_t seed, uint64_t n, int mode, int bits, double *bad_a, double *bad_b) {
  uint64_t s = seed, bad = 0;
  const uint64_t lim = 1ull << bits;
  for (uint64_t i = 0; i < n; i++) {
    uint64_t b, a;
    if (mode == 0) { b = 1 + mix(s) % (lim - 1); a = mix(s) % b; }
    else if (mode == 1) {
      const int k = 1 + (int)(mix(s) % (unsigned)bits);
      switch (mix(s) % 4) {
        case 0: b = (1ull << k) - 1; break;
        case 1: b = (1ull << k) + 1; break;
        case 2: b = ((mix(s) % 4096) | 1) << (mix(s) % (unsigned)(bits > 12 ? bits - 12 : 1)); break;
        default: b = lim - 1 - mix(s) % 64; break;
      }
      if (b < 2) b = 2;
      if (b >= lim) b = lim - 1;
      const int j = 1 + (int)(mix(s) % 53);
      const unsigned __int128 t = (unsigned __int128)b * (mix(s) % (1ull << (j > 20 ? 20 : j)));
      a = (uint64_t)(t >> (j > 20 ? 20 : j));
      const int64_t d = (int64_t)(mix(s) % 5) - 2;
      if ((int64_t)a + d >= 0) a = (uint64_t)((int64_t)a + d);
      a %= b;
    } else { b = 1 + mix(s) % 1000; a = mix(s) % b; }
    const double q = div_small_f64((double)a, (double)b, 1.0 / (double)b), want = (double)a / (double)b;
    if (!(q == want)) { if (bad == 0) { *bad_a = (double)a; *bad_b = (double)b; } bad++; }
  }
  return bad;
}
// scorer inputs as clusters have them: capacities = cores x 1000 / GiB, requests = sums of round pod requests; also adversarial small integers
uint64_t eh_score_mismatches(uint64_t seed, uint64_t n, int mode, int64_t *bad) {
  uint64_t s = seed, nb = 0;
  static const int64_t cpus[] = {100, 250, 500, 1000, 1500, 2000, 3000, 4000, 8000}, mems[] = {64, 128, 256, 512, 1000, 1024, 2048, 3000, 4096, 8192};
  for (uint64_t i = 0; i < n; i++) {
    int64_t ac, am, nc, nm, tc, tm;
    if (mode == 0) {
      ac = (int64_t)(1 + mix(s) % 128) * 1000; am = (int64_t)(1 + mix(s) % 512) << 30;
      nc = 0; nm = 0;
      for (int k = (int)(mix(s) % 40); k > 0; k--) { nc += cpus[mix(s) % 9]; nm += mems[mix(s) % 10] << 20; }
      tc = cpus[mix(s) % 9]; tm = mems[mix(s) % 10] << 20;
    } else if (mode == 1) {   // small integers: every boundary of the 0..10 scores and of the balanced fraction is hit
      ac = (int64_t)(mix(s) % 40); am = (int64_t)(mix(s) % 40); nc = (int64_t)(mix(s) % 45); nm = (int64_t)(mix(s) % 45); tc = (int64_t)(mix(s) % 5); tm = (int64_t)(mix(s) % 5);
    } else {                  // up to the 2^48 envelope
      ac = (int64_t)(mix(s) % ((1ull << 48) - 1)); am = (int64_t)(mix(s) % ((1ull << 48) - 1));
      nc = (int64_t)(mix(s) % (uint64_t)(ac + 2)); nm = (int64_t)(mix(s) % (uint64_t)(am + 2)); tc = (int64_t)(mix(s) % 100000); tm = (int64_t)(mix(s) % (1ull << 34));
      if (nc + tc >= (1ll << 48) || nm + tm >= (1ll << 48)) continue;
    }
    const int wL = (int)(mix(s) % 4), wM = (int)(mix(s) % 6), wB = (int)(mix(s) % 3);
    const uint32_t got = eh_score(tc, tm, nc, nm, ac, am, wL, wM, wB), want = ref_score(tc, tm, nc, nm, ac, am, wL, wM, wB);
    if (got != want) { if (nb == 0) { bad[0] = tc; bad[1] = tm; bad[2] = nc; bad[3] = nm; bad[4] = ac; bad[5] = am; bad[6] = wL; bad[7] = wM; bad[8] = wB; } nb++; }
  }
  return nb;
}
uint64_t eh_le_mismatches(uint64_t seed, uint64_t n) {
  uint64_t s = seed, bad = 0;
  static const double epss[] = {EPS_CPU, EPS_MEM, EPS_SCALAR};
  for (uint64_t i = 0; i < n; i++) {
    const double eps = epss[mix(s) % 3];
    double r = (double)(mix(s) % (1ull << 40)) * ((mix(s) & 1) ? 1.0 : 0.001), l;
    switch (mix(s) % 4) {
      case 0: l = (double)(mix(s) % (1ull << 40)); break;
      case 1: l = r + eps * (((double)(mix(s) % 2001) - 1000.0) / 1000.0); break;          // around the epsilon
      case 2: l = nextafter(r + eps, (mix(s) & 1) ? 1e300 : -1e300); break;
      default: l = r - (double)(mix(s) % 3) * eps; break;
    }
    if ((mix(s) % 16) == 0) r = -r;   // Idle can have drifted below zero (sub-epsilon requests)
    if (le_eps(l, r, eps) != ref_le(l, r, eps)) bad++;
  }
  return bad;
}
}
