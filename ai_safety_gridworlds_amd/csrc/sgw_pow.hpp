// sgw_pow.hpp -- pow() as the host C library computes it (glibc >= 2.28, sysdeps/ieee754/dbl-64/e_pow.c, the FMA build that
// x86-64 machines with FMA dispatch to), restated for the GPU so that resource regrowth matches the reference's
// math.pow bit for bit (island_navigation_ex.py:603-636, island_navigation_ex_ma.py:771-781, aintelope_savanna.py:1251-1254).
//
// glibc's pow is NOT correctly rounded (error bound ~0.52 ulp): about 1 result in 100 differs from the correctly rounded
// value, so neither a correctly rounded device pow nor the device math library's pow can stand in for it.  The algorithm:
//   log(x) = k ln2 + log(c) + log1p(z/c - 1) as hi + lo (table of 128 c's, degree-7 polynomial, exact r = fma(z, 1/c, -1))
//   y log(x) as ehi + elo (fma), exp(ehi + elo) = 2^(k/128) (1 + tail + expm1(r)) (table of 128 scales, degree-5 polynomial)
// with every a*b+c contracted to an fma exactly where GCC contracts glibc's source (validated: 0 mismatches in 2*10^7
// inputs against pow() on the build host, tests/test_pow.py; on the device, tests/test_pow_gpu.py).
// Domain (asserted by the callers' arithmetic, not checked here): x finite, positive, normal; y finite; result normal and
// |y log x| in [2^-54, 2^9) -- regrowth uses x in [2, 61], y = 1.1.
// Tables: csrc/sgw_pow_tables.inc, generated from the build host's libm by tools/gen_pow_tables.py.
#pragma once
#include <stdint.h>

#include <math.h>
#include <string.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SGW_POW_FN __host__ __device__ inline
#define SGW_POW_DEV_FN __device__ inline
#define SGW_POW_TABLE __device__ const
#else
#define SGW_POW_FN static inline
#define SGW_POW_DEV_FN static inline
#define SGW_POW_TABLE static const
#endif

#include "sgw_pow_tables.inc"

SGW_POW_FN double sgw_pow_asd(unsigned long long u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)u);
#else
  double x; memcpy(&x, &u, 8); return x;
#endif
}
SGW_POW_FN unsigned long long sgw_pow_asu(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (unsigned long long)__double_as_longlong(x);
#else
  unsigned long long u; memcpy(&u, &x, 8); return u;
#endif
}

// `log_tab` / `exp_tab`: SGW_POW_LOG_TAB / SGW_POW_EXP_TAB or a copy of them (a kernel may stage the 5 KB in LDS: the two
// lookups are data-dependent per lane, and an LDS read costs a fraction of an L2 round trip)
// `LH` / `EH`: the two 9- / 8-entry constant heads (SGW_POW_LOG_HEAD / SGW_POW_EXP_HEAD or the host copy sgw_create checks the
// running libm against)
SGW_POW_FN double sgw_glibc_pow_core(double x, double y, const unsigned long long* LH, const unsigned long long* EH,
                                     const unsigned long long* log_tab, const unsigned long long* exp_tab) {
  const unsigned long long OFF = 0x3fe6955500000000ULL;
  // ---- log_inline (e_pow.c:60-140)
  const unsigned long long ix = sgw_pow_asu(x);
  const unsigned long long tmp = ix - OFF;
  const int i = (int)((tmp >> (52 - 7)) & 127);
  const int k = (int)((long long)tmp >> 52);
  const unsigned long long iz = ix - (tmp & (0xfffULL << 52));
  const double z = sgw_pow_asd(iz), kd = (double)k;
  const double invc = sgw_pow_asd(log_tab[3 * i]), logc = sgw_pow_asd(log_tab[3 * i + 1]),
               logctail = sgw_pow_asd(log_tab[3 * i + 2]);
  const double ln2hi = sgw_pow_asd(LH[0]), ln2lo = sgw_pow_asd(LH[1]);
  const double A0 = sgw_pow_asd(LH[2]), A1 = sgw_pow_asd(LH[3]), A2 = sgw_pow_asd(LH[4]), A3 = sgw_pow_asd(LH[5]),
               A4 = sgw_pow_asd(LH[6]), A5 = sgw_pow_asd(LH[7]), A6 = sgw_pow_asd(LH[8]);
  const double r = fma(z, invc, -1.0);
  const double t1 = fma(kd, ln2hi, logc);
  const double t2 = t1 + r;
  const double lo1 = fma(kd, ln2lo, logctail);
  const double lo2 = t1 - t2 + r;
  const double ar = A0 * r, ar2 = r * ar, ar3 = r * ar2;
  double hi = t2 + ar2;
  const double lo3 = fma(ar, r, -ar2);
  const double lo4 = t2 - hi + ar2;
  const double p = ar3 * fma(ar2, fma(ar2, fma(r, A6, A5), fma(r, A4, A3)), fma(r, A2, A1));
  double lo = lo1 + lo2 + lo3 + lo4 + p;
  const double yl = hi + lo;
  lo = hi - yl + lo; hi = yl;
  // ---- y * log(x) (e_pow.c:360-375) and exp_inline (e_pow.c:240-300), sign_bias = 0
  const double ehi = y * hi;
  const double elo = fma(y, lo, fma(y, hi, -ehi));
  const double invln2N = sgw_pow_asd(EH[0]), shift = sgw_pow_asd(EH[1]), negln2hiN = sgw_pow_asd(EH[2]), negln2loN = sgw_pow_asd(EH[3]);
  const double C2 = sgw_pow_asd(EH[4]), C3 = sgw_pow_asd(EH[5]), C4 = sgw_pow_asd(EH[6]), C5 = sgw_pow_asd(EH[7]);
  const double zz = invln2N * ehi;
  double kd2 = zz + shift;
  const unsigned long long ki = sgw_pow_asu(kd2);
  kd2 -= shift;
  double rr = fma(kd2, negln2loN, fma(kd2, negln2hiN, ehi));
  rr += elo;
  const unsigned long long idx = 2 * (ki & 127);
  const unsigned long long top = ki << (52 - 7);
  const double tl = sgw_pow_asd(exp_tab[idx]);
  const unsigned long long sbits = exp_tab[idx + 1] + top;
  const double r2 = rr * rr;
  const double tm = fma(r2 * r2, fma(rr, C5, C4), fma(r2, fma(rr, C3, C2), tl + rr));
  const double scale = sgw_pow_asd(sbits);
  return fma(scale, tm, scale);
}
SGW_POW_DEV_FN double sgw_glibc_pow_t(double x, double y, const unsigned long long* log_tab, const unsigned long long* exp_tab) {
  return sgw_glibc_pow_core(x, y, SGW_POW_LOG_HEAD, SGW_POW_EXP_HEAD, log_tab, exp_tab);
}
SGW_POW_DEV_FN double sgw_glibc_pow(double x, double y) { return sgw_glibc_pow_t(x, y, SGW_POW_LOG_TAB, SGW_POW_EXP_TAB); }

#if defined(__HIPCC__)
// The workgroup (64 or 256 threads) copies both tables (3 KB + 2 KB) into `lds` (16-byte aligned), all L2 hits.  Every
// thread must call it (k_engine's init_ctx is such a place); the reader passes `lds` and `lds + 128 * 3` to
// sgw_glibc_pow_t.
constexpr int SGW_POW_LDS_BYTES = (128 * 3 + 256) * 8;
// 320 pieces of 16 bytes (192 log + 128 exp) over THREADS threads: ceil(320 / THREADS) unconditional loads per thread, piece
// (t + j * THREADS) mod 320 (the wrap-around rewrites a few pieces with the same bytes; no branch for the loads to sink into)
template <int THREADS> struct SgwPowStageT { static constexpr int NP = (320 + THREADS - 1) / THREADS; uint4 v[NP]; };
using SgwPowStage = SgwPowStageT<256>;
template <int THREADS>
__device__ inline int sgw_pow_piece(int j) {       // a workgroup with more threads than THREADS repeats the same pieces
  const int i = (int)(threadIdx.x & (THREADS - 1)) + j * THREADS;
  return i >= 320 ? i - 320 : i;
}
template <int THREADS>
__device__ inline void sgw_pow_stage_issue(SgwPowStageT<THREADS>& st) {               // loads only, unconditional
  const uint4* lg = reinterpret_cast<const uint4*>(SGW_POW_LOG_TAB);     // 192 x 16 B
  const uint4* ex = reinterpret_cast<const uint4*>(SGW_POW_EXP_TAB);     // 128 x 16 B
#pragma unroll
  for (int j = 0; j < SgwPowStageT<THREADS>::NP; ++j) {
    const int i = sgw_pow_piece<THREADS>(j);
    const uint4* src = i < 192 ? lg + i : ex + (i - 192);
    st.v[j] = *src;
  }
}
template <int THREADS>
__device__ inline void sgw_pow_stage_commit(const SgwPowStageT<THREADS>& st, void* lds) {
  uint4* dst = reinterpret_cast<uint4*>(lds);
#pragma unroll
  for (int j = 0; j < SgwPowStageT<THREADS>::NP; ++j) dst[sgw_pow_piece<THREADS>(j)] = st.v[j];
}
__device__ inline double sgw_glibc_pow_lds(double x, double y, const void* lds) {
  const unsigned long long* t = reinterpret_cast<const unsigned long long*>(lds);
  return sgw_glibc_pow_t(x, y, t, t + 128 * 3);
}
#endif

