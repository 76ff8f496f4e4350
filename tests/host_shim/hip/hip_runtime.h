// TEST INFRASTRUCTURE ONLY: a minimal stand-in for <hip/hip_runtime.h> so that the per-lane device code of the game
// families (csrc/sgw_*.hpp: State load/store, begin_episode, play, board, metrics) also compiles for the HOST, where
// gcc's sanitizers (-fsanitize=undefined,address) and pattern-initialised locals can look at it.  One "lane" runs at a
// time; nothing here is a CPU path of the product (libsgw.so has none) -- tests/test_host_families.py is the only user.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline
#define __launch_bounds__(x)
#define __shared__

struct sgw_dim3 { unsigned x, y, z; };
inline thread_local sgw_dim3 threadIdx{0, 0, 0}, blockIdx{0, 0, 0}, blockDim{64, 1, 1}, gridDim{1, 1, 1};

struct uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return uint4{a, b, c, d}; }
struct double2 { double x, y; };
struct float4 { float x, y, z, w; };
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }

inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xffffffu) * (b & 0xffffffu); }
inline long long __double_as_longlong(double x) { long long u; std::memcpy(&u, &x, 8); return u; }
inline double __longlong_as_double(long long u) { double x; std::memcpy(&x, &u, 8); return x; }
inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p |= v; return o; }
inline unsigned long long __ballot(bool b) { return b ? 1ull : 0ull; }
inline int __any(int p) { return p != 0; }
template <class T> inline T __shfl_xor(T v, int, int) { return v; }
inline void __syncthreads() {}
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
