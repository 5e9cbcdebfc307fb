// kb_res.hpp — api.Resource (pkg/scheduler/api/resource_info.go) as the engine's host side computes with it: a dense vector with the map-presence
// semantics of ScalarResources.  Host code (kb_host.hpp includes this) and, since round 3, device code: the proportion water-fill kernel
// (kb_waterfill.hip) runs the same functions, so that its arithmetic is this text and not a second restatement.
#pragma once
#include <stdint.h>

#include <cmath>

#include "../../include/kb_engine.h"

#if defined(__HIP__)
#include <hip/hip_runtime.h>
#define KB_HD __host__ __device__
#else
#define KB_HD
#endif

namespace kb {

constexpr double kMinMilliCPU = 10.0;              // pkg/scheduler/api/resource_info.go:68
constexpr double kMinMilliScalar = 10.0;           // :69
constexpr double kMinMemory = 10.0 * 1024 * 1024;  // :70

// api.Resource (resource_info.go:28-38): dense vector; `mask` bit (d-2) <=> ScalarResources has key d; mask==0 <=> nil map
struct Res {
  double v[KB_MAX_RES];
  uint32_t mask;
  KB_HD Res() : mask(0) { for (unsigned i = 0; i < KB_MAX_RES; i++) v[i] = 0.0; }
  KB_HD bool has(int d) const { return (mask >> (d - 2)) & 1u; }
  KB_HD void setk(int d) { mask |= 1u << (d - 2); }
  KB_HD double get(int d) const { return d < 2 ? v[d] : (has(d) ? v[d] : 0.0); }   // resource_info.go:349-361
};

KB_HD inline bool le_func(double l, double r, double diff) { return l < r || fabs(l - r) < diff; }

KB_HD inline void res_add(Res &r, const Res &rr, int R) {   // resource_info.go:128-140
  r.v[0] += rr.v[0];
  r.v[1] += rr.v[1];
  for (int d = 2; d < R; d++)
    if (rr.has(d)) { r.setk(d); r.v[d] += rr.v[d]; }
}
KB_HD inline bool res_less_equal(const Res &r, const Res &rr, int R) {   // resource_info.go:268-302
  if (!le_func(r.v[0], rr.v[0], kMinMilliCPU)) return false;
  if (!le_func(r.v[1], rr.v[1], kMinMemory)) return false;
  if (r.mask == 0) return true;
  for (int d = 2; d < R; d++) {
    if (!r.has(d) || r.v[d] <= kMinMilliScalar) continue;
    if (rr.mask == 0) return false;
    if (!le_func(r.v[d], rr.get(d), kMinMilliScalar)) return false;
  }
  return true;
}
KB_HD inline bool res_sub(Res &r, const Res &rr, int R) {   // resource_info.go:143-160; false where Go panics
  if (!res_less_equal(rr, r, R)) return false;
  r.v[0] -= rr.v[0];
  r.v[1] -= rr.v[1];
  for (int d = 2; d < R; d++) {
    if (!rr.has(d)) continue;
    if (r.mask == 0) return true;
    r.setk(d);
    r.v[d] -= rr.v[d];
  }
  return true;
}
KB_HD inline void res_multi(Res &r, double ratio, int R) {   // resource_info.go:217-224
  r.v[0] = r.v[0] * ratio;
  r.v[1] = r.v[1] * ratio;
  for (int d = 2; d < R; d++)
    if (r.has(d)) r.v[d] = r.v[d] * ratio;
}
KB_HD inline bool res_less(const Res &r, const Res &rr, int R) {   // resource_info.go:227-265
  if (!(r.v[0] < rr.v[0])) return false;
  if (!(r.v[1] < rr.v[1])) return false;
  if (r.mask == 0) {
    if (rr.mask != 0)
      for (int d = 2; d < R; d++)
        if (rr.has(d) && rr.v[d] <= kMinMilliScalar) return false;
    return true;
  }
  if (rr.mask == 0) return false;
  for (int d = 2; d < R; d++)
    if (r.has(d) && !(r.v[d] < rr.get(d))) return false;
  return true;
}
KB_HD inline bool res_is_empty(const Res &r, int R) {   // resource_info.go:93-105
  if (!(r.v[0] < kMinMilliCPU && r.v[1] < kMinMemory)) return false;
  for (int d = 2; d < R; d++)
    if (r.has(d) && r.v[d] >= kMinMilliScalar) return false;
  return true;
}
KB_HD inline void res_diff(const Res &r, const Res &rr, Res &inc, Res &dec, int R) {   // resource_info.go:305-337
  inc = Res();
  dec = Res();
  if (r.v[0] > rr.v[0]) inc.v[0] += r.v[0] - rr.v[0]; else dec.v[0] += rr.v[0] - r.v[0];
  if (r.v[1] > rr.v[1]) inc.v[1] += r.v[1] - rr.v[1]; else dec.v[1] += rr.v[1] - r.v[1];
  for (int d = 2; d < R; d++) {
    if (!r.has(d)) continue;
    double rq = rr.get(d);
    if (r.v[d] > rq) { inc.setk(d); inc.v[d] += r.v[d] - rq; }
    else { dec.setk(d); dec.v[d] += rq - r.v[d]; }
  }
}
KB_HD inline Res helpers_min(const Res &l, const Res &r, int R) {   // api/helpers/helpers.go:28-44
  Res res;
  res.v[0] = fmin(l.v[0], r.v[0]);
  res.v[1] = fmin(l.v[1], r.v[1]);
  if (l.mask == 0 || r.mask == 0) return res;
  for (int d = 2; d < R; d++)
    if (l.has(d)) { res.setk(d); res.v[d] = fmin(l.v[d], r.get(d)); }
  return res;
}
KB_HD inline double helpers_share(double l, double r) {   // api/helpers/helpers.go:47-60
  if (r == 0) return l == 0 ? 0.0 : 1.0;
  return l / r;
}

}  // namespace kb
