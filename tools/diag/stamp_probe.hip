// Diagnostic build ONLY (never shipped, never timed for a quoted number): per-wave s_memtime stamps around the phases of
// the island step kernel, to see where a launch's ~10 us go.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DSGW_STAMPS -I. tools/diag/stamp_probe.hip -o /tmp/stamp_probe && /tmp/stamp_probe
// It drives k_engine<IslandPacked, K_STEP> directly (same code as libsgw.so, compiled with stamps) on 65 536 envs.
// -DSGW_ISLAND_EW=n sets the env-waves per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define SGW_STAMP_DECL unsigned long long* sgw_stamps;
#include "../../ai_safety_gridworlds_amd/csrc/sgw_island.hpp"
#include "../../ai_safety_gridworlds_amd/csrc/sgw_kernels.hpp"

using namespace sgw;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 65536;
  const int NW = (int)(n / 64);
  using Fam = IslandPacked;
  constexpr int EW = env_waves<Fam, K_STEP>();
  const int K = 10, HW = 48, words = Fam::words(K);
  KArgs a; memset(&a, 0, sizeof(a));
  KSpec& sp = a.sp;
  sp.family = 0; sp.H = 6; sp.W = 8; sp.HW = HW; sp.K = K; sp.M = 9; sp.A = 1; sp.max_iterations = 100; sp.flags = 1 | 4;
  sp.action_lo = 0; sp.n_actions = 5; sp.words = words; sp.start_cell[0] = 18; kspec_derive(sp);
  const int slots[12] = {0, 1, 2, 3, -1, 4, 5, 6, 7, 8, 9, -1};
  for (int ag = 0; ag < SGW_MAX_AGENTS; ++ag) for (int u = 0; u < SGW_MAX_K; ++u) sp.dim_slot[ag][u] = (ag == 0 && u < 12) ? slots[u] : -1;
  for (int m = 0; m < SGW_MAX_M; ++m) sp.metric_slot[m] = m < 9 ? m : -1;
  const char* art = "WW######WW  D  WWSA W  WW  W  GWW  F  WWW#######";
  std::vector<unsigned char> tables(TABLE_BYTES, 0);
  for (int i = 0; i < HW; ++i) { tables[i] = art[i] == 'A' ? ' ' : art[i]; tables[SGW_MAX_CELLS + i] = art[i]; tables[2 * SGW_MAX_CELLS + i] = 1; }
  double params[SGW_N_PARAMS] = {-1, 50, -1, -1, 20, 20, 0, 0, 0, 0, 0, 0, 40, 30, -50, -50, -1, -1, 0, 10, -1, -20, 4, 0, 10, -1, -20, 4,
                                 1.1, 20, 20, 1.1, 20, 20};
  memcpy(tables.data() + 3 * SGW_MAX_CELLS + 512, params, sizeof(params));
  unsigned char* d_tables; unsigned long long* d_state; signed char* d_act; unsigned long long* d_stamps;
  unsigned char* d_board; double* d_reward; unsigned char* d_st; unsigned char* d_term; int* d_safety; int* d_frame;
  CK(hipMalloc(&d_tables, TABLE_BYTES)); CK(hipMemcpy(d_tables, tables.data(), TABLE_BYTES, hipMemcpyHostToDevice));
  const size_t sw = (size_t)state_alloc_words(words, n);
  CK(hipMalloc(&d_state, sw * 8));
  std::vector<unsigned long long> st0(sw, 0ull);
  for (long long i = 0; i < n; ++i) st0[state_index(0, i, words)] = (3ull << 32) | (15ull << 36);
  CK(hipMemcpy(d_state, st0.data(), sw * 8, hipMemcpyHostToDevice));
  const int T = 200;
  CK(hipMalloc(&d_act, T * n));
  std::vector<signed char> acts(T * n);
  unsigned x = 12345; for (auto& v : acts) { x = x * 1664525u + 1013904223u; v = (signed char)((x >> 24) % 5); }
  CK(hipMemcpy(d_act, acts.data(), T * n, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_stamps, (size_t)NW * 8 * 8)); CK(hipMalloc(&d_board, n * HW)); CK(hipMalloc(&d_reward, n * K * 8));
  CK(hipMalloc(&d_st, n)); CK(hipMalloc(&d_term, n)); CK(hipMalloc(&d_safety, n * 4)); CK(hipMalloc(&d_frame, n * 4));
  a.tables = d_tables; a.state = reinterpret_cast<uint64_t*>(d_state); a.n_pad = n; a.n_envs = n; a.mode = MODE_STEP; a.T = 1;
  const int omask = argc > 3 ? atoi(argv[3]) : 63;      // bit0 board, 1 reward, 2 step_type, 3 term_reason, 4 safety, 5 frame
  if (omask & 1) a.out.board = d_board;
  if (omask & 2) a.out.reward = d_reward;
  if (omask & 4) a.out.step_type = d_st;
  if (omask & 8) a.out.term_reason = d_term;
  if (omask & 16) a.out.safety = d_safety;
  if (omask & 32) a.out.frame = d_frame;
  a.sgw_stamps = d_stamps;
  a.lp = lds_plan(HW, 1, K, 9, 1, lds_need(a, false), 0); a.need = lds_need(a, false);
  const size_t lds = lds_total_bytes(HW, 1, K, 9, 1, lds_need(a, false), 0, Fam::LDS_EXTRA, EW, 1) + (argc > 2 ? atoll(argv[2]) : 0);
  printf("n %lld, %d env-waves per workgroup, dynamic LDS %zu bytes per workgroup\n", n, EW, lds);
  std::vector<unsigned long long> h((size_t)NW * 8);
  std::vector<double> starts, ends;
  double acc[8] = {0}; int cnt = 0;
  for (int t = 0; t < T; ++t) {
    a.actions = d_act + t * n;
    hipLaunchKernelGGL((k_engine<Fam, K_STEP>), dim3((NW + EW - 1) / EW), dim3(EW * 64), lds, 0, SGW_HOT_ARGS(a), a);
    if (t >= 100 && t % 10 == 0) {
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), d_stamps, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0min = ~0ull, tend = 0;
      for (int w = 0; w < NW; ++w) { t0min = std::min(t0min, h[w * 8]); tend = std::max(tend, h[w * 8 + 5]); }
      for (int k = 1; k < 6; ++k) { double s = 0; for (int w = 0; w < NW; ++w) s += (double)(h[w * 8 + k] - h[w * 8 + k - 1]); acc[k] += s / NW; }
      double s0 = 0; for (int w = 0; w < NW; ++w) s0 += (double)(h[w * 8] - t0min); acc[0] += s0 / NW;
      acc[6] += (double)(tend - t0min); ++cnt;
      unsigned long long r0 = ~0ull;                       // s_memrealtime: 100 MHz, one clock for all XCDs
      for (int w = 0; w < NW; ++w) r0 = std::min(r0, h[w * 8 + 6]);
      for (int w = 0; w < NW; ++w) { starts.push_back((double)(h[w * 8 + 6] - r0) * 0.01); ends.push_back((double)(h[w * 8 + 7] - r0) * 0.01); }
    }
  }
  CK(hipDeviceSynchronize());
  const char* names[] = {"wave start skew (mean start - first start)", "issue loads -> state+tables arrived", "play (rules)",
                         "return staging + output phase issued", "accumulate tail", "state stores issued", "first wave start -> last wave end"};
  printf("mean shader cycles per wave (s_memtime ticks; 2.4 GHz nominal)\n");
  // rows 0 and 6 compare s_memtime values of DIFFERENT waves: the shader clock is per XCD, so they mean nothing across the chip --
  // the launch-wide spread is the s_memrealtime percentile lines below
  for (int k = 1; k < 6; ++k) printf("  %-48s %9.0f cycles  %6.2f us\n", names[k], acc[k] / cnt, acc[k] / cnt / 2400.0);
  std::sort(starts.begin(), starts.end()); std::sort(ends.begin(), ends.end());
  auto pct = [](const std::vector<double>& v, double q) { return v[(size_t)(q * (v.size() - 1))]; };
  printf("wave START after the launch's first wave (us, s_memrealtime): p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f\n",
         pct(starts, .1), pct(starts, .5), pct(starts, .9), pct(starts, .99), starts.back());
  printf("wave END   after the launch's first wave (us):                p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f\n",
         pct(ends, .1), pct(ends, .5), pct(ends, .9), pct(ends, .99), ends.back());
  return 0;
}
