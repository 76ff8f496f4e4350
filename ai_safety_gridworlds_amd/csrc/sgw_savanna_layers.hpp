// sgw_savanna_layers.hpp -- aintelope_savanna's unoccluded observation layers from the state bitmaps (sgw_state_layers), on the
// plane writer of sgw_kernels.hpp.  Include after sgw_savanna.hpp and sgw_kernels.hpp.
#pragma once

#include "sgw_kernels.hpp"
#include "sgw_savanna.hpp"

namespace sgw {

// Unoccluded observation layers straight from the state bitmaps (the rendered board only shows the top drape of a cell):
// layers[n][l][cell] for the characters in layer_chars ('#', ' ', W P D F d f G S, '0', '1'); the gap layer is set only
// where every other layer is blank when gap_only_blank (observe_gaps_only_where_other_layers_are_blank=True, SV:1690).
// A workgroup takes `epb` envs: their 27 bitmap words + the position word go to LDS, phase 1 reduces a cell to a 12-bit code
// vector (bit k = "source k is on here": wall, W, G, S, P, D, F, d, f, agent 0, agent 1, gap), phase 2 is the plane writer of
// sgw_kernels.hpp -- four cells per lane, one dword store per layer, the layer's source picked by a scalar shift.  (Round 2:
// a thread per cell re-reading the state words from global memory and storing a byte per layer: 137 us at 65 536 envs.)
constexpr int SAV_LAYER_WORDS = 28;
__global__ __launch_bounds__(PLANES_THREADS) void k_savanna_layers(const uint64_t* state, long long n_pad, long long n, int words, PlaneGeom g, int W, int two,
                                                                   const uint8_t* layer_chars, int gap_only_blank, int epb, uint8_t* layers) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pl_lds[];
  const int HW = g.HW, L = g.P;
  const long long env0 = (long long)blockIdx.x * epb;
  const int n_env = n - env0 < epb ? (int)(n - env0) : epb;
  uint64_t* sw = reinterpret_cast<uint64_t*>(pl_lds);                      // [epb][28]: word 1, then words W_STATIC .. W_DYN + 14
  uint32_t* bits = reinterpret_cast<uint32_t*>(sw + epb * SAV_LAYER_WORDS);   // [epb * HW] code vectors
  uint32_t* code = bits + epb * HW;                                         // [L]: which source a layer shows (12 = none)
  for (int i = threadIdx.x; i < n_env * SAV_LAYER_WORDS; i += PLANES_THREADS) {
    const int e = i / SAV_LAYER_WORDS, k = i - e * SAV_LAYER_WORDS;
    sw[i] = state[state_index(k == 0 ? 1 : Savanna::W_STATIC + k - 1, env0 + e, words)];
  }
  for (int l = threadIdx.x; l < L; l += PLANES_THREADS) {
    uint32_t c = 12u;
    switch (layer_chars[l]) {
      case '#': c = 0; break;  case 'W': c = 1; break;  case 'G': c = 2; break;  case 'S': c = 3; break;  case 'P': c = 4; break;  case 'D': c = 5; break;
      case 'F': c = 6; break;  case 'd': c = 7; break;  case 'f': c = 8; break;  case '0': c = 9; break;  case '1': c = 10; break; case ' ': c = 11; break;
    }
    code[l] = c;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n_env * HW; i += PLANES_THREADS) {
    const int e = (int)div_recip((uint32_t)i, (uint32_t)g.recip_HW), cell = i - e * HW;
    const uint64_t* w = sw + e * SAV_LAYER_WORDS;
    const int wi = cell >> 6, sh = cell & 63;
    uint32_t b = 0u;
#pragma unroll
    for (int k = 0; k < 9; ++k) b |= (uint32_t)((w[1 + 3 * k + wi] >> sh) & 1ull) << k;      // wall W G S | P D F d f: three words each
    const uint64_t w1 = w[0];
    const int c0 = (int)(w1 & 0xff) * W + (int)((w1 >> 8) & 0xff), c1 = (int)((w1 >> 16) & 0xff) * W + (int)((w1 >> 24) & 0xff);
    b |= (cell == c0 ? 1u : 0u) << 9;
    b |= ((two && cell == c1) ? 1u : 0u) << 10;
    b |= ((gap_only_blank ? b == 0u : (b & 1u) == 0u) ? 1u : 0u) << 11;
    bits[i] = b;
  }
  __syncthreads();
  planes_expand4(layers + env0 * L * HW, g, n_env, [&](int e, int c, int nvalid, auto put) {
    const uint32_t* mp = bits + e * HW + c;
    const uint32_t m0 = mp[0], m1 = mp[nvalid > 1 ? 1 : 0], m2 = mp[nvalid > 2 ? 2 : 0], m3 = mp[nvalid > 3 ? 3 : 0];
    for (int p = 0; p < L; ++p) {
      const uint32_t k = code[p];                                           // scalar
      put(p, ((m0 >> k) & 1u) | (((m1 >> k) & 1u) << 8) | (((m2 >> k) & 1u) << 16) | (((m3 >> k) & 1u) << 24));
    }
  });
}

}  // namespace sgw
