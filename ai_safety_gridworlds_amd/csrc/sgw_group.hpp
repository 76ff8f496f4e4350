// sgw_group.hpp -- k_engine_group: one grid over several engines of different families (see GroupArgs in sgw_kernels.hpp).
// Include after every family header.
#pragma once

#include "sgw_kernels.hpp"

namespace sgw {

// the member families: single-agent, four env-waves per step workgroup, pipelined fused rollout -- every workgroup of the
// grid has GROUP_THREADS threads whatever its family.  (The round kernels -- firemaker, island_navigation_ex_ma, savanna -- fill
// the chip on their own and differ in workgroup shape.)
#define SGW_GROUP_FAMILIES(X)                                                                                          \
  X(TAG_ISLAND_PACKED, IslandPacked) X(TAG_ISLAND, Island) X(TAG_BOAT, Boat) X(TAG_SAFEINT, SafeInt) X(TAG_TILE, Tile) \
  X(TAG_SOKOBAN, Sokoban) X(TAG_CONVEYOR, Conveyor) X(TAG_TOMATO, Tomato) X(TAG_FRIEND_FOE, FriendFoe) X(TAG_WHISKY, Whisky) X(TAG_ROCKS, Rocks)

// leading scalar arguments (preloaded into SGPRs at wave launch): which member a workgroup belongs to is known without a
// memory access, so the member's KArgs are the kernel's FIRST scalar loads and not its second, dependent round trip
// (a graph replay's kernarg segment is in device memory: a scalar-cache miss is most of a microsecond)
struct GroupHot { int n, fb1, fb2, fb3, tags; };              // tags: 8 bits per member
#define SGW_GROUP_HOT_ARGS(h) (h).n, (h).fb1, (h).fb2, (h).fb3, (h).tags, 0
constexpr unsigned GROUP_ARGS_OFFSET = 24;                    // 6 ints, then the 8-byte aligned GroupArgs
struct GroupKernargMirror { int n, fb1, fb2, fb3, tags, pad; GroupArgs g; };
static_assert(offsetof(GroupKernargMirror, g) == GROUP_ARGS_OFFSET, "GROUP_ARGS_OFFSET does not match k_engine_group's leading arguments");

template <int KIND>
__global__ __launch_bounds__(GROUP_THREADS) void k_engine_group(int hot_n, int hot_fb1, int hot_fb2, int hot_fb3, int hot_tags, int hot_pad,
                                                                 const GroupArgs g_in) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const KArgs __attribute__((address_space(4))) * KArgsSeg;
  const int b = (int)blockIdx.x;
  int m = 0;
  m = (hot_n > 1 && b >= hot_fb1) ? 1 : m; m = (hot_n > 2 && b >= hot_fb2) ? 2 : m; m = (hot_n > 3 && b >= hot_fb3) ? 3 : m;
  const int first = m == 0 ? 0 : (m == 1 ? hot_fb1 : (m == 2 ? hot_fb2 : hot_fb3));
  const int tag = (hot_tags >> (8 * m)) & 0xff;
  const long long block = b - first;
  const unsigned off = GROUP_ARGS_OFFSET + (unsigned)(offsetof(GroupArgs, a) + (size_t)m * sizeof(KArgs));
  KArgs a;
  {
    // read through a pointer the compiler cannot see through: a member index that is only known at run time must not turn
    // the by-value argument block into a private copy
    KArgsSeg seg = (KArgsSeg)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + off);
    // touch every cache line of the member's argument block at once: the bodies read their fields in half a dozen batches with
    // a wait after each (SGPR pressure), and a graph replay's kernarg segment is cold -- six serial misses were 1.8 us per launch
    static_assert(sizeof(KArgs) <= 704 && sizeof(KArgs) >= 580, "the touch loads below cover 11 lines");
    {
      uint32_t t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10;
      asm volatile("s_load_dword %0, %11, 0x0\n\ts_load_dword %1, %11, 0x40\n\ts_load_dword %2, %11, 0x80\n\ts_load_dword %3, %11, 0xc0\n\t"
                   "s_load_dword %4, %11, 0x100\n\ts_load_dword %5, %11, 0x140\n\ts_load_dword %6, %11, 0x180\n\ts_load_dword %7, %11, 0x1c0\n\t"
                   "s_load_dword %8, %11, 0x200\n\ts_load_dword %9, %11, 0x240\n\ts_load_dword %10, %11, %12\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(t0), "=&s"(t1), "=&s"(t2), "=&s"(t3), "=&s"(t4), "=&s"(t5), "=&s"(t6), "=&s"(t7), "=&s"(t8), "=&s"(t9), "=&s"(t10)
                   : "s"(seg), "i"((int)sizeof(KArgs) - 4) : "memory");
    }
    asm volatile("" : "+s"(seg) : : "memory");
    a = *seg;
  }
  switch (tag) {                                             // scalar: one family body per workgroup
#define SGW_GROUP_CASE(TAG, F)                                                                              \
    case TAG:                                                                                               \
      static_assert(group_member<F, KIND>(), "group members share the workgroup shape");                   \
      engine_body<F, KIND>(a, block, off);                                                                  \
      break;
    SGW_GROUP_FAMILIES(SGW_GROUP_CASE)
#undef SGW_GROUP_CASE
    default: break;
  }
#endif
}

}  // namespace sgw
