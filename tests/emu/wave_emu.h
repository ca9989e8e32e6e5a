// wave_emu.h — TEST-ONLY single-wavefront SIMT emulator for the CPU.
//
// Lets tests/ compile gym_solo_amd/csrc/solo_step_kernel.h UNCHANGED with g++ (optionally with
// -fsanitize=address,undefined, which the GPU pool cannot run) and execute it lane by lane:
// the 64 lanes of a workgroup are ucontext fibres that are switched at every cross-lane
// operation (wave_sync / wave_readlane / wave_shfl_xor / wave_ballot).  The kernel keeps all
// cross-lane operations in wave-uniform control flow, which is also what the hardware needs.
// Nothing in the product path includes this file.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

// Minimal x86-64 fibre switch (callee-saved registers + stack pointer).  glibc's swapcontext
// makes a sigprocmask syscall per switch, ~50x slower for the ~10^5 switches of one env step.
extern "C" void solo_emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl solo_emu_switch
.type solo_emu_switch,@function
solo_emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size solo_emu_switch, .-solo_emu_switch
)");

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)

struct EmuDim3 { int x = 0, y = 0, z = 0; };
static EmuDim3 threadIdx, blockIdx, blockDim, gridDim;

namespace solo {

class WaveEmu {
 public:
  static constexpr int W = 64;
  static WaveEmu& get() { static WaveEmu e; return e; }

  // runs `body` once per lane of workgroup `block`
  void run_block(int block, int grid, const std::function<void()>& body) {
    body_ = &body;
    blockIdx.x = block; gridDim.x = grid; blockDim.x = W;
    if (stacks_.empty()) stacks_.resize((size_t)W * kStack + 64);
    int remaining = W;
    for (int l = 0; l < W; ++l) {
      uintptr_t top = (uintptr_t)(stacks_.data() + (size_t)(l + 1) * kStack);
      top &= ~(uintptr_t)15;
      void** sp = (void**)top;
      *--sp = nullptr;                       // fake return address of the trampoline
      *--sp = (void*)&WaveEmu::trampoline;   // popped by `ret` in solo_emu_switch
      for (int r = 0; r < 6; ++r) *--sp = nullptr;
      sp_[l] = (void*)sp;
      finished_[l] = false;
    }
    while (remaining > 0) {
      for (int l = 0; l < W; ++l) {
        if (finished_[l]) continue;
        cur_ = l;
        threadIdx.x = l;
        solo_emu_switch(&sched_sp_, sp_[l]);
        if (finished_[l]) --remaining;
      }
    }
  }
  void yield() {
    const int me = cur_;
    solo_emu_switch(&sp_[me], sched_sp_);
  }
  uint64_t exchange(uint64_t mine, int src) {
    xchg_[cur_] = mine;
    yield();
    const uint64_t v = xchg_[src & (W - 1)];
    yield();
    return v;
  }
  int lane() const { return cur_; }
  uint64_t peek(int l) const { return xchg_[l]; }
  void post(uint64_t v) { xchg_[cur_] = v; }

 private:
  static constexpr size_t kStack = 512 * 1024;
  static void trampoline() {
    WaveEmu& e = get();
    (*e.body_)();
    e.finished_[e.cur_] = true;
    for (;;) e.yield();  // never returns: the scheduler does not resume finished lanes
  }
  void* sched_sp_ = nullptr;
  void* sp_[W];
  std::vector<char> stacks_;
  bool finished_[W];
  int cur_ = 0;
  uint64_t xchg_[W];
  const std::function<void()>* body_ = nullptr;
};

inline int lane_id() { return WaveEmu::get().lane(); }
inline int block_id() { return blockIdx.x; }
inline void wave_sync() { WaveEmu::get().yield(); }
template <typename P> inline P* wave_opaque(P* p) { return p; }
inline int wave_opaque_lane(int lane) { return lane; }
inline unsigned long long wave_opaque_bits(unsigned long long x) { return x; }
inline int wave_fresh_lane() { return WaveEmu::get().lane(); }
inline int wave_uniform(int x) { return x; }
inline int wave_readlane_int(int x, int lane) { return (int)(uint32_t)WaveEmu::get().exchange((uint32_t)x, lane); }
inline void wave_set_priority_level(int) {}
template <typename B> using ColdArgs = const B*;
template <typename B> inline ColdArgs<B> wave_cold_args(const B& by_value) { return &by_value; }
inline int wave_slot_id() { return 0; }
// the robot-migration queue's device-scope primitives: plain memory operations (the emulated waves run one after the other)
inline int wave_atomic_add(int32_t* p, int v) { const int o = *p; *p = o + v; return o; }
inline int wave_atomic_load(const int32_t* p) { return *p; }
inline void wave_atomic_store(int32_t* p, int v) { *p = v; }
template <typename T> inline T wave_load_shared(const T* p) { return *p; }
template <typename T> inline void wave_store_shared(T* p, T v) { *p = v; }
inline void wave_release_device() {}
inline void wave_acquire_device() {}
inline void wave_backoff() {}
inline void wave_fault_set(int32_t* p) { *p = 1; }
inline int wave_xcc_id() { return (int)(blockIdx.x & 7); }  // (as the hardware deals workgroups over the XCDs)
inline void wave_fence_global() {}  // (one emulated wave runs its lanes as fibres over plain memory)
template <typename T> struct RowDot {
  T g[6], h[2];
  void set(const T* gh, const T* hh) { for (int i = 0; i < 6; ++i) g[i] = gh[i]; h[0] = hh[0]; h[1] = hh[1]; }
  T dot(const T* rg, const T* rh) const {
    const T a1 = g[0] * rg[0] + g[1] * rg[1] + g[2] * rg[2];
    const T a2 = g[3] * rg[3] + g[4] * rg[4] + g[5] * rg[5];
    return (a1 + a2) + (h[0] * rh[0] + h[1] * rh[1]);
  }
  T dot(const T* rg, const T* rh, T same) const {  // (slot space: the joint-space part counts between rows of one leg)
    const T a1 = g[0] * rg[0] + g[1] * rg[1] + g[2] * rg[2];
    const T a2 = g[3] * rg[3] + g[4] * rg[4] + g[5] * rg[5];
    return (a1 + a2) + same * (h[0] * rh[0] + h[1] * rh[1]);
  }
};

// the register-resident matrix of the GPU build (solo_wave_ops.h ColumnBank<T>) as a plain array.  Like the GPU
// build: f32 with lane = row, f64 in SLOT space with 32 resident column slots (the overflow path evaluates column()).
template <typename T> struct ColumnBank {
  static constexpr bool kResident = true;
  static constexpr bool kCompact = sizeof(T) == 8;
  static constexpr int kSlots = kCompact ? 32 : 64, kRowStride = 8;
  static constexpr int kBanks = 1;
  static constexpr unsigned long long bank_lanes(int) { return ~0ull; }
  RowDot<T> own;
  T nid;
  int lane;
  const T* rowvec;
  const T* hext = nullptr;                 // lane = row (f32): the joint-space parts by leg slot
  const unsigned char* rowleg = nullptr;   // slot space (f64): the leg of every slot's row, and this lane's
  int leg = 0;
  T a[64];
  void init(const T* gh, const T* hh, T nid_, int lane_, const T* rowvec_, const T* hext_) {
    own.set(gh, hh); nid = nid_; lane = lane_; rowvec = rowvec_; hext = hext_;
    for (int r = 0; r < 64; ++r) a[r] = std::nan("");  // a column that was never built must never be used
  }
  void init(const T* gh, const T* hh, T nid_, int lane_, const T* rowvec_, const unsigned char* rowleg_, int leg_) {
    own.set(gh, hh); nid = nid_; lane = lane_; rowvec = rowvec_; rowleg = rowleg_; leg = leg_;
    for (int r = 0; r < 64; ++r) a[r] = std::nan("");
  }
  T column(int r) const {
    const T m = (lane == r) ? T(0) : nid;
    if (kCompact) return m * own.dot(rowvec + kRowStride * r, rowvec + kRowStride * r + 6, rowleg[r] == leg ? T(1) : T(0));
    return m * own.dot(rowvec + kRowStride * r, hext + 8 * r);
  }
  void build(int r) { a[r] = column(r); }
  // (the pipelined build of the slot-space kernel: solo_wave_ops.h)
  struct Row { T g[6], h[2]; int leg; };
  Row fetch(int r) const {
    Row x;
    for (int i = 0; i < 6; ++i) x.g[i] = rowvec[kRowStride * r + i];
    x.h[0] = rowvec[kRowStride * r + 6]; x.h[1] = rowvec[kRowStride * r + 7];
    x.leg = rowleg[r];
    return x;
  }
  void build_from(int r, const Row& x) { const T m = (lane == r) ? T(0) : nid; a[r] = m * own.dot(x.g, x.h, x.leg == leg ? T(1) : T(0)); }
  T get(int, int r) const { return a[r]; }
};

inline float wave_readlane(float x, int lane) {
  uint32_t b; std::memcpy(&b, &x, 4);
  b = (uint32_t)WaveEmu::get().exchange(b, lane);
  std::memcpy(&x, &b, 4); return x;
}
inline double wave_readlane(double x, int lane) {
  uint64_t b; std::memcpy(&b, &x, 8);
  b = WaveEmu::get().exchange(b, lane);
  std::memcpy(&x, &b, 8); return x;
}
template <typename T> inline T emu_shfl_xor(T x, int mask) { return wave_readlane(x, lane_id() ^ mask); }
template <typename T> inline T wave_sum_legs(T x) {
  x += emu_shfl_xor(x, 16);
  x += emu_shfl_xor(x, 32);
  return x;
}
template <typename T> inline T wave_other_half16(T x) { return emu_shfl_xor(x, 8); }
template <typename T> inline T wave_from_lower_half16(T x) { const T o = emu_shfl_xor(x, 8); return (lane_id() & 8) ? x : o; }
template <typename T> inline T wave_from_upper_half16(T x) { const T o = emu_shfl_xor(x, 8); return (lane_id() & 8) ? o : x; }
template <int N, typename T> inline T wave_lane_below(T x) {
  const int l = lane_id();
  const T y = wave_readlane(x, (l & 15) >= N ? l - N : l);
  return (l & 15) >= N ? y : T(0);
}
// the value of lane - N of the whole wave (lanes < N get 0)
template <int N, typename T> inline T wave_slot_below(T x) {
  const int l = lane_id();
  const T y = wave_readlane(x, l >= N ? l - N : l);
  return l >= N ? y : T(0);
}
inline int wave_count_below(unsigned long long mask) { return __builtin_popcountll(mask & ((1ull << lane_id()) - 1ull)); }
// PUSH: lane l's value goes to lane dst[l] (a permutation); PULL: lane l gets the value of lane src[l]
inline uint64_t emu_push_bits(uint64_t bits, int dst) {
  WaveEmu& e = WaveEmu::get();
  e.post((uint64_t)(unsigned)dst);
  e.yield();
  int src = -1;
  for (int l = 0; l < WaveEmu::W; ++l) if ((int)e.peek(l) == e.lane()) { if (src >= 0) { std::fprintf(stderr, "wave_push: not a permutation\n"); std::abort(); } src = l; }
  if (src < 0) { std::fprintf(stderr, "wave_push: not a permutation\n"); std::abort(); }
  e.yield();
  e.post(bits);
  e.yield();
  const uint64_t v = e.peek(src);
  e.yield();
  return v;
}
inline int wave_push_int(int x, int dst) { return (int)(uint32_t)emu_push_bits((uint32_t)x, dst); }
inline float wave_push(float x, int dst) { uint32_t b; std::memcpy(&b, &x, 4); b = (uint32_t)emu_push_bits(b, dst); std::memcpy(&x, &b, 4); return x; }
inline double wave_push(double x, int dst) { uint64_t b; std::memcpy(&b, &x, 8); b = emu_push_bits(b, dst); std::memcpy(&x, &b, 8); return x; }
template <typename T> inline T wave_pull(T x, int src) { return wave_readlane(x, src); }
template <typename T> inline T wave_sum_group16(T x) {
  x += emu_shfl_xor(x, 8);
  x += emu_shfl_xor(x, 4);
  x += emu_shfl_xor(x, 2);
  x += emu_shfl_xor(x, 1);
  return x;
}
template <typename T> inline T wave_sum_all(T x) { return wave_sum_legs(wave_sum_group16(x)); }
template <typename T> inline void wave_reduce_rows(T (&z)[6], T (&y)[2]) {
  for (int i = 0; i < 6; ++i) z[i] = wave_sum_all(z[i]);
  y[0] = wave_sum_group16(y[0]);
  y[1] = wave_sum_group16(y[1]);
}
// (the same sums through LDS - solo_wave_ops.h: wave_reduce_rows_lds, its association restated)
constexpr int kReduceScratch = 64 * 9;
template <typename T> inline void wave_reduce_rows_lds(T (&z)[6], T (&y)[2], T* scratch, int lane) {
  for (int i = 0; i < 6; ++i) scratch[lane * 9 + i] = z[i];
  scratch[lane * 9 + 6] = y[0]; scratch[lane * 9 + 7] = y[1];
  wave_sync();
  const T* col = scratch + (lane >> 3) * 72 + (lane & 7);
  T p = col[0];
  for (int i = 1; i < 8; ++i) p += col[9 * i];
  p += emu_shfl_xor(p, 8);
  const T total = wave_sum_legs(p);
  for (int i = 0; i < 6; ++i) z[i] = wave_readlane(total, i);
  y[0] = wave_readlane(p, (lane & 48) + 6);
  y[1] = wave_readlane(p, (lane & 48) + 7);
  wave_sync();  // (fibres: nobody rewrites the scratch while another lane still reads it)
}
inline unsigned long long wave_ballot(bool p) {
  WaveEmu& e = WaveEmu::get();
  e.post(p ? 1 : 0);
  e.yield();
  unsigned long long m = 0;
  for (int l = 0; l < WaveEmu::W; ++l) m |= (unsigned long long)(e.peek(l) & 1) << l;
  e.yield();
  return m;
}

template <typename T> inline T wave_math_table(int) { return T(0); }
template <typename T> struct Real;
template <> struct Real<float> {
  static float sqrt(float x) { return std::sqrt(x); }
  static float rsqrt(float x) { return 1.0f / std::sqrt(x); }
  static float rcp(float x) { return 1.0f / x; }
  static constexpr int kTabSize = 0;
  static void sincos(float x, float* s, float* c, const float* = nullptr) { *s = std::sin(x); *c = std::cos(x); }
  static void sinc_cos(float x2, float* sinc, float* c, const float* = nullptr) { const float x = std::sqrt(x2); *c = std::cos(x); *sinc = x > 1e-12f ? std::sin(x) / x : 1.0f; }
  static float atan2(float y, float x) { return std::atan2(y, x); }
  static float cos_of_asin(float x) { return std::sqrt(std::fmax(0.0f, std::fma(-x, x, 1.0f))); }
  static float asin(float x) { return std::asin(x); }
  static float exp(float x) { return std::exp(x); }
  static float abs(float x) { return std::fabs(x); }
  static float min(float a, float b) { return std::fmin(a, b); }
  static float max(float a, float b) { return std::fmax(a, b); }
  static float clamp(float x, float lo, float hi) { return std::fmin(std::fmax(x, lo), hi); }
  static bool finite(float x) { return std::isfinite(x); }
  static float fma(float a, float b, float c) { return std::fma(a, b, c); }
  static float floor(float x) { return std::floor(x); }
  static float big() { return 3.0e38f; }
  static float half_pi() { return 1.57079637f; }
  static float half_ulp() { return 5.9604645e-8f; }
};
template <> struct Real<double> {
  static double sqrt(double x) { return std::sqrt(x); }
  static double rsqrt(double x) { return 1.0 / std::sqrt(x); }
  static double rcp(double x) { return 1.0 / x; }
  static constexpr int kTabSize = 0;
  static void sincos(double x, double* s, double* c, const double* = nullptr) { *s = std::sin(x); *c = std::cos(x); }
  static void sinc_cos(double x2, double* sinc, double* c, const double* = nullptr) { const double x = std::sqrt(x2); *c = std::cos(x); *sinc = x > 1e-12 ? std::sin(x) / x : 1.0; }
  static double atan2(double y, double x) { return std::atan2(y, x); }
  static double cos_of_asin(double x) { return std::sqrt((1.0 - x) * (1.0 + x)); }
  static double asin(double x) { return std::asin(x); }
  static double exp(double x) { return std::exp(x); }
  static double abs(double x) { return std::fabs(x); }
  static double min(double a, double b) { return std::fmin(a, b); }
  static double max(double a, double b) { return std::fmax(a, b); }
  static double clamp(double x, double lo, double hi) { return std::fmin(std::fmax(x, lo), hi); }
  static bool finite(double x) { return std::isfinite(x); }
  static double fma(double a, double b, double c) { return std::fma(a, b, c); }
  static double floor(double x) { return std::floor(x); }
  static double big() { return 1.0e300; }
  static double half_pi() { return 1.5707963267948966; }
  static double half_ulp() { return 1.1102230246251565e-16; }
};

inline void stats_add(double* p, double v) { *p += v; }

}  // namespace solo
