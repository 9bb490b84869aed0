// The three covariance functions of gpsa/util/util.py:8-66 as one device function (shared by kmat.hip - the covariance
// matrices and their backward - and proj64.hip, which forms K_uf on the fly).
#pragma once
#include "common.hpp"

namespace gpsa {

// k and the two derivative coefficients:  dk/dz_d = cd * (z_d - x_d),  pl = dk/d ls_u
template <typename T, int KIND>
__device__ __forceinline__ void cov_eval(const T* z, const T* x, int D, T ell, T inv_ell, T var,
                                         T& k, T& cd, T& pl) {
  if (KIND == GPSA_K_RBF) {
    T s = T(0);
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
      if (d < D) {
        T u = (z[d] - x[d]) * inv_ell;  // reference divides by the lengthscale before squaring
        s += u * u;
      }
    k = var * t_exp<T>(T(-0.5) * s);
    cd = -k * inv_ell * inv_ell;
    pl = k * s;
  } else {
    T s = T(0);
#pragma unroll
    for (int d = 0; d < MAXD; ++d)
      if (d < D) {
        T u = z[d] - x[d];
        s += u * u;
      }
    T dist = t_sqrt<T>(s + T(1e-10));  // eps inside the sqrt (util.py:44-45, 61-62)
    if (KIND == GPSA_K_MATERN12) {
      k = var * t_exp<T>(T(-0.5) * dist * inv_ell);  // reference's non-standard 0.5 factor
      cd = T(-0.5) * k * inv_ell / dist;
      pl = T(0.5) * k * dist * inv_ell;
    } else {
      T zz = T(1.7320508075688772) * dist * inv_ell;
      T e = var * t_exp<T>(-zz);
      k = (T(1) + zz) * e;
      cd = T(-3) * e * inv_ell * inv_ell;
      pl = e * zz * zz;
    }
  }
}

}  // namespace gpsa
