// ekf_hip_rt.h -- hand-written HIP runtime shared by every generated rednose_amd filter library.
//
// Plays the role rednose/templates/ekf_c.c plays in the reference (the fixed text spliced into every
// generated filter, /root/reference/rednose/helpers/ekf_sym.py:207-208), but for gfx950:
//   * wave-private LDS transposition between the AoS batch layout in HBM (x:(N,D), P:(N,E,E),
//     z:(N,Z) row-major, the natural numpy/torch batch of the reference's per-filter buffers) and a
//     lane-per-filter register layout, with fully coalesced 16-byte global accesses;
//   * in-register square-root-free Cholesky (L D L^T) factor / solve for the Z x Z innovation covariance
//     (S is SPD; the reference uses fullPivLu, ekf_c.c:89,101 -- same solution up to rounding);
//   * host-side plumbing for the C ABI: error reporting, scratch buffers for the single-filter
//     host-pointer entry points, launch geometry.
// Written for CDNA4 only: 64-lane wavefronts, one wavefront per workgroup, no portability layer.
#pragma once

#include <hip/hip_runtime.h>
#include <cstring>
#include <mutex>
#include <utility>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

namespace rn {

constexpr int WAVE = 64;

// ------------------------------------------------------------------------------------------------
// status / error reporting (the reference's entry points are void and assert; ours never abort)
// ------------------------------------------------------------------------------------------------
enum Status : int {
  OK = 0,
  ERR_HIP = 1,         // a HIP runtime call failed (no device, launch failure, ...)
  ERR_ARG = 2,         // null / negative / unknown-kind argument
  ERR_ALIGN = 3,       // device pointer not 16-byte aligned
  ERR_UNSUPPORTED = 4, // entry point not generated for this model (e.g. fused run above 32 error states)
};

struct ErrorState {
  int code = 0;
  int hip_code = 0;
  char msg[256] = {0};
};

inline ErrorState& err() {
  static thread_local ErrorState e;
  return e;
}

inline int fail(int code, int hip_code, const char* what, int line) {
  ErrorState& e = err();
  e.code = code;
  e.hip_code = hip_code;
  snprintf(e.msg, sizeof(e.msg), "rednose_amd: %s failed at line %d (status %d, hip %d: %s)", what, line, code, hip_code,
           hip_code ? hipGetErrorString((hipError_t)hip_code) : "-");
  fprintf(stderr, "%s\n", e.msg);
  return code;
}

#define RN_HIP(expr)                                                             \
  do {                                                                           \
    hipError_t rn_e_ = (expr);                                                   \
    if (rn_e_ != hipSuccess) return rn::fail(rn::ERR_HIP, (int)rn_e_, #expr, __LINE__); \
  } while (0)

#define RN_REQUIRE(cond, code)                                                   \
  do {                                                                           \
    if (!(cond)) return rn::fail((code), 0, #cond, __LINE__);                    \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Launch geometry: one wavefront (64 filters, lane-per-filter) per workgroup.  The dispatcher places
// workgroup b on XCD b % 8, so the tile -> workgroup map is kept identical for every kernel of a
// library (tile = blockIdx.x + k * gridDim.x with gridDim.x a multiple of 8): a filter's x/P written by
// one step are still in the SAME XCD's L2 when the next launch reads them.
constexpr int MAX_GRID = 256 * 16;   // 16 single-wave workgroups per CU, then grid-stride

inline int grid_for_tiles(int64_t tiles) {
  int64_t g = tiles < MAX_GRID ? tiles : MAX_GRID;
  if (g >= 8) g -= g % 8;
  return (int)(g < 1 ? 1 : g);
}

// ------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------

// Orders this wavefront's LDS traffic.  The LDS services one wave's instructions in issue order, so
// no s_waitcnt / s_barrier is needed -- only the compiler must not move accesses across this point.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Rendezvous of a workgroup's wavefronts that exchange data through LDS ONLY (the two-wavefront fused run, codegen/emit_run2.py):
// this wavefront's LDS traffic has completed, then s_barrier.  Deliberately not __syncthreads(): its workgroup-scope release also
// waits for vmcnt(0), i.e. for every store the wavefront has in flight -- the fused run's covariance trace (32 KB per step) would
// be drained at every barrier.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// One-way hand-off between two wavefronts of a workgroup through a word of LDS: the producer's earlier LDS accesses are performed before
// the flag's store, the consumer polls and its later LDS accesses are performed after the load that saw the flag.  Both wavefronts are
// resident (same workgroup), so the wait cannot deadlock.  Release / acquire at WORKGROUP scope restricted to the LDS address space (the
// "local" argument of the fence): under the memory model that is what orders the hand-off for the other wavefront -- a relaxed atomic with
// wavefront-scope fences, which this was, relied on the in-order LDS queue and on hipcc not moving LDS accesses across the flag -- and it
// costs what the hardware needs anyway, `s_waitcnt lgkmcnt(0)` in front of the flag's store: a release over ALL address spaces would also
// wait for vmcnt(0), i.e. drain the covariance trace stores in flight, like __syncthreads() (see wg_barrier).
__device__ __forceinline__ void flag_set(int* flag, int v) {
  wave_lds_sync();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  if ((threadIdx.x & 63) == 0) __hip_atomic_store(flag, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void flag_wait(int* flag, int v) {
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != v) __builtin_amdgcn_s_sleep(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
  wave_lds_sync();
}

// Forces `v` to exist in registers at this point of the instruction stream (an empty volatile asm that "modifies" it):
// arithmetic producing v cannot sink below, arithmetic consuming it cannot rise above.  No instruction is emitted.
__device__ __forceinline__ void pin(double& v) { asm volatile("" : "+v"(v)); }

// LDS image of a tile: filter f's record starts at f * lds_stride<EPF>() doubles.  An ODD stride makes the
// lane-per-filter ds_read_b64 / ds_write_b64 accesses conflict-free (the linear layout costs 4-way conflicts on P,
// 864 cycles per wave in the round-1 PMC run), but the index arithmetic of the padded copy cost more than the
// conflicts it removed (measured: 10.7 -> 11.3 us per launch, 252 -> 256+ VGPRs), so the linear layout is the
// default; -DRN_LDS_PAD=1 selects the padded one.  With the linear stride the div/mod below folds away.
#ifndef RN_LDS_PAD
#define RN_LDS_PAD 0
#endif
template <int EPF>
__device__ __forceinline__ constexpr int lds_stride() { return RN_LDS_PAD ? (EPF | 1) : EPF; }

// Copy one tile (cnt <= 64 filters x EPF doubles, contiguous in HBM) into this wave's LDS image.
// Full tiles move as 16-byte vectors, lane l taking vectors l, l+64, ... (1 KiB per wave-instruction).
template <int EPF>
__device__ __forceinline__ void tile_g2l(const double* __restrict__ g, int cnt, double* lds, int lane) {
  constexpr int NV = 32 * EPF;                      // double2 vectors in a full tile
  constexpr int STR = lds_stride<EPF>();
  if (cnt == WAVE) {
    const double2* __restrict__ g2 = reinterpret_cast<const double2*>(g);
    double2 v[(NV + WAVE - 1) / WAVE];
#pragma unroll
    for (int i = 0; i < (NV + WAVE - 1) / WAVE; i++) {
      const int idx = lane + i * WAVE;
      v[i] = g2[((NV % WAVE == 0) || idx < NV) ? idx : NV - 1];       // unconditional, clamped (see copy_g2l)
    }
#pragma unroll
    for (int i = 0; i < (NV + WAVE - 1) / WAVE; i++) {
      const int idx = lane + i * WAVE;
      if ((NV % WAVE == 0) || idx < NV) {
#if RN_LDS_PAD
        const int e = 2 * idx;
        const int f = e / EPF, k = e - f * EPF;
        lds[f * STR + k] = v[i].x;
        if (k + 1 < EPF) lds[f * STR + k + 1] = v[i].y; else lds[(f + 1) * STR] = v[i].y;
#else
        reinterpret_cast<double2*>(lds)[idx] = v[i];
#endif
      }
    }
  } else {
    const int total = cnt * EPF;
    for (int e = lane; e < total; e += WAVE) {
      const int f = e / EPF, k = e - f * EPF;
      lds[f * STR + k] = g[e];
    }
  }
}

template <int EPF>
__device__ __forceinline__ void tile_l2g(double* __restrict__ g, int cnt, const double* lds, int lane) {
  constexpr int NV = 32 * EPF;
  constexpr int STR = lds_stride<EPF>();
  if (cnt == WAVE) {
    double2* __restrict__ g2 = reinterpret_cast<double2*>(g);
#pragma unroll
    for (int i = 0; i < (NV + WAVE - 1) / WAVE; i++) {
      const int idx = lane + i * WAVE;
      if ((NV % WAVE == 0) || idx < NV) {
#if RN_LDS_PAD
        const int e = 2 * idx;
        const int f = e / EPF, k = e - f * EPF;
        double2 v;
        v.x = lds[f * STR + k];
        v.y = (k + 1 < EPF) ? lds[f * STR + k + 1] : lds[(f + 1) * STR];
        g2[idx] = v;
#else
        g2[idx] = reinterpret_cast<const double2*>(lds)[idx];
#endif
      }
    }
  } else {
    const int total = cnt * EPF;
    for (int e = lane; e < total; e += WAVE) {
      const int f = e / EPF, k = e - f * EPF;
      g[e] = lds[f * STR + k];
    }
  }
}

// Register-staged prefetch of one tile (cnt <= 64 filters x EPF doubles): issue() starts the coalesced global loads (lane l
// takes doubles l, l + 64, ...), commit() puts them into the LDS image later.  ONE representation and unconditional, clamped
// loads on purpose: with a 16-byte path for full tiles next to an 8-byte path for ragged ones the staging arrays were
// assigned under different conditions, hipcc kept the struct in scratch memory, and every step of the fused run then went
// through 48 bytes of scratch per lane (a scratch access costs a wavefront that is alone on its SIMD about a microsecond).
template <int EPF>
struct TilePrefetch {
  static constexpr int STR = RN_LDS_PAD ? (EPF | 1) : EPF;
  double s[EPF];
  __device__ __forceinline__ void issue(const double* __restrict__ g, int cnt, int lane) {
    const int last = cnt * EPF - 1;
#pragma unroll
    for (int i = 0; i < EPF; i++) {
      const int idx = lane + i * WAVE;
      s[i] = g[idx <= last ? idx : last];
    }
  }
  __device__ __forceinline__ void commit(double* lds, int cnt, int lane) const {
#pragma unroll
    for (int i = 0; i < EPF; i++) {
      const int e = lane + i * WAVE;
      if (e < cnt * EPF) {
        const int f = e / EPF, k = e - f * EPF;
        lds[f * STR + k] = s[i];
      }
    }
  }
};

// Copy `nd` contiguous doubles (nd <= MAXD, source 16-byte aligned) between HBM and this wave's LDS as
// 16-byte vectors, all loads issued before the first LDS store so they overlap.
template <int MAXD>
__device__ __forceinline__ void copy_g2l(const double* __restrict__ g, int nd, double* lds, int lane) {
  constexpr int IT = (MAXD / 2 + WAVE - 1) / WAVE;
  const int nv = nd >> 1;
  const double2* __restrict__ g2 = reinterpret_cast<const double2*>(g);
  double2* l2 = reinterpret_cast<double2*>(lds);
  double2 v[IT];
  // Loads are UNCONDITIONAL on a clamped index: a predicated `if (idx < nv) v[i] = ...` demotes v[] to scratch and
  // hipcc then waits for every load before the next one (load, s_waitcnt vmcnt(0), scratch_store, ...), which
  // serialised the whole HBM latency IT times per copy (61 % of the live kernel's cycles in the round-1 PMC run).
  const int last = nv > 0 ? nv - 1 : 0;
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int idx = lane + i * WAVE;
    v[i] = g2[idx < nv ? idx : last];
  }
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int idx = lane + i * WAVE;
    if (idx < nv) l2[idx] = v[i];
  }
  if ((nd & 1) && lane == 0) lds[nd - 1] = g[nd - 1];
}

// NT = true: nontemporal stores, for data that is written once and read much later (the filtered trace of a fused run: tens of
// gigabytes between its write and the smoother's read) -- it should not displace the working set from the L2.
template <int MAXD, bool NT = false>
__device__ __forceinline__ void copy_l2g(double* __restrict__ g, int nd, const double* lds, int lane) {
  typedef double v2d_ __attribute__((ext_vector_type(2)));
  constexpr int NV = MAXD / 2;
  constexpr int IT = (NV + WAVE - 1) / WAVE;
  constexpr int FULL = NV / WAVE;            // iterations in which every lane has an element of a full buffer
  constexpr int BATCH = 8;                   // LDS reads in flight before the first store of a batch (32 registers)
  const int nv = nd >> 1;
  v2d_* __restrict__ g2 = reinterpret_cast<v2d_*>(g);
  const v2d_* l2 = reinterpret_cast<const v2d_*>(lds);
  if (nd == MAXD) {
    // A full buffer (every tile but a ragged last one): no per-element predicate, BATCH reads issued before their stores.  The guarded
    // loop below compiles to compare - exec mask - ds_read - s_waitcnt lgkmcnt(0) - store per KiB, i.e. one exposed LDS latency per
    // iteration: 1.9 us for the 31 KiB of a fused-run trace step (profiles/tuning_notes.md, round 5).
#pragma unroll
    for (int b = 0; b < FULL; b += BATCH) {
      v2d_ v[BATCH];
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        if (b + i < FULL) v[i] = l2[lane + (b + i) * WAVE];
      }
#pragma unroll
      for (int i = 0; i < BATCH; i++) {
        if (b + i < FULL) {
          if constexpr (NT) __builtin_nontemporal_store(v[i], g2 + lane + (b + i) * WAVE);
          else g2[lane + (b + i) * WAVE] = v[i];
        }
      }
      wave_lds_sync();
    }
    if constexpr (FULL < IT) {
      const int idx = lane + FULL * WAVE;
      if (idx < NV) {
        if constexpr (NT) __builtin_nontemporal_store(l2[idx], g2 + idx);
        else g2[idx] = l2[idx];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < IT; i++) {
      const int idx = lane + i * WAVE;
      if (idx < nv) {
        if constexpr (NT) __builtin_nontemporal_store(l2[idx], g2 + idx);
        else g2[idx] = l2[idx];
      }
    }
  }
  if ((nd & 1) && lane == 0) g[nd - 1] = lds[nd - 1];
}

// A generic pointer into LDS carries the LDS byte offset in its low 32 bits; building the address_space(3) pointer
// from that integer avoids the generic->local addrspacecast (its null check trips an hipcc 7.2 backend assertion,
// "V_CMP_NE_U32 $src_shared_base: incorrect register class", when the pointer comes through a function argument).
typedef __attribute__((address_space(3))) void* lds_void_ptr;
__device__ __forceinline__ lds_void_ptr lds_offset_ptr(const double* p) {
  return (lds_void_ptr)(uint32_t)(uintptr_t)p;
}

// Asynchronous HBM -> LDS copy of `nd` contiguous doubles (16-byte aligned source, LDS image linear) with
// global_load_lds_dwordx4: no VGPR staging, the data lands in LDS while the wave keeps computing.  Each
// wave-instruction writes one 1 KiB stripe: LDS address = uniform base + lane * 16.  The consumer must call
// async_wait() (s_waitcnt vmcnt(0)) and then wave_lds_sync() before reading the image.
template <int MAXD>
__device__ __forceinline__ void async_copy_g2l(const double* __restrict__ g, int nd, double* lds, int lane) {
  constexpr int IT = (MAXD / 2 + WAVE - 1) / WAVE;
  const int nv = nd >> 1;
#pragma unroll
  for (int i = 0; i < IT; i++) {
    const int idx = lane + i * WAVE;
    if (idx < nv) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2 * idx), lds_offset_ptr(lds + 2 * i * WAVE), 16, 0, 0);
    }
  }
  if ((nd & 1) && lane == 0) lds[nd - 1] = g[nd - 1];
}

// Records that may start on an odd double (8-byte but not 16-byte aligned): the LDS image is shifted by `sh` = 0/1
// doubles so that record element i lives at lds[sh + i] and every 16-byte transfer is aligned on both sides.  The
// load starts one double early (the previous record's last element, always inside the same allocation); the store
// writes the odd head and tail elements as scalars and never touches a neighbour's element.
__device__ __forceinline__ int odd_start(const double* g) {
  return (int)(((uintptr_t)g >> 3) & 1);
}

template <int MAXD>
__device__ __forceinline__ void async_copy_g2l_any(const double* __restrict__ g, int nd, double* lds, int lane) {
  const int sh = odd_start(g);
  async_copy_g2l<MAXD>(g - sh, nd + sh, lds, lane);
}

template <int MAXD>
__device__ __forceinline__ void copy_l2g_any(double* __restrict__ g, int nd, const double* lds, int sh, int lane) {
  if (sh && lane == 0 && nd > 0) g[0] = lds[1];
  copy_l2g<MAXD>(g + sh, nd - sh, lds + 2 * sh, lane);
}

__device__ __forceinline__ void async_wait() {
  __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0), expcnt / lgkmcnt untouched
}

// tile_g2l with direct HBM -> LDS transfers for full tiles (no VGPR staging, no ds_write); the caller must
// async_wait() before the wave_lds_sync() that precedes the first read of the image.
template <int EPF>
__device__ __forceinline__ void tile_g2l_async(const double* __restrict__ g, int cnt, double* lds, int lane) {
#if RN_LDS_PAD
  tile_g2l<EPF>(g, cnt, lds, lane);
#else
  constexpr int NV = 32 * EPF;
  if (cnt == WAVE) {
#pragma unroll
    for (int i = 0; i < (NV + WAVE - 1) / WAVE; i++) {
      const int idx = lane + i * WAVE;
      if ((NV % WAVE == 0) || idx < NV) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + 2 * idx), lds_offset_ptr(lds + 2 * i * WAVE), 16, 0, 0);
      }
    }
  } else {
    tile_g2l<EPF>(g, cnt, lds, lane);
  }
#endif
}

// lane-per-filter register <-> LDS (filter `lane` owns lds[lane*STR .. lane*STR+EPF), STR = lds_stride<EPF>())
template <int EPF>
__device__ __forceinline__ void lds_to_regs(const double* lds, int lane, double (&r)[EPF]) {
#pragma unroll
  for (int k = 0; k < EPF; k++) r[k] = lds[lane * lds_stride<EPF>() + k];
}

template <int EPF>
__device__ __forceinline__ void regs_to_lds(double* lds, int lane, const double (&r)[EPF]) {
#pragma unroll
  for (int k = 0; k < EPF; k++) lds[lane * lds_stride<EPF>() + k] = r[k];
}

// Value held by the other lane of an (even, odd) lane pair: two DPP moves (quad_perm [1,0,3,2]), no LDS traffic.
__device__ __forceinline__ double pair_xchg(const double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Reciprocal for the factorisation below: v_rcp_f64 seed + two Newton steps (about 1 ulp).  The IEEE division
// sequence hipcc emits for 1.0/d is ~10 DEPENDENT instructions; on this hardware a dependent fp64 instruction costs
// ~40 cycles for a lone wavefront (tools/fp64_ilp.hip: 40.7 cycles per FMA with one chain, 10.8 with eight), and the
// factorisation sits on the critical path of every update.  No denormal / overflow scaling: d is a variance.
__device__ __forceinline__ double fast_recip(const double d) {
#ifdef RN_EXACT_MATH      // tuning knob exact_math=1 (codegen/tuning.py): IEEE division / square root and the library's sin, cos everywhere -- the
  return 1.0 / d;         // build tests/test_gpu_live.py measures the fast primitives against
#endif
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

// Compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N - 1>{}).  Unlike
// `#pragma unroll`, the body may contain scheduling barriers (hipcc does not duplicate those when unrolling, which turns
// register arrays indexed by the loop variable into scratch).
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// 1 / sqrt(a) for a > 0: v_rsq_f64 seed + two Newton steps (same reasoning as fast_recip: sqrt followed by an IEEE
// division is ~100 dependent instructions per pivot of a Cholesky factorisation).
__device__ __forceinline__ double fast_rsqrt(const double a) {
#ifdef RN_EXACT_MATH
  return 1.0 / sqrt(a);
#endif
  double y = __builtin_amdgcn_rsq(a);
  const double h = 0.5 * a;
  y = fma(y, fma(-h * y, y, 0.5), y);
  y = fma(y, fma(-h * y, y, 0.5), y);
  return y;
}

// fast_recip / fast_rsqrt for the MODEL'S OWN expressions (rednose_amd/codegen/lower.py prints reciprocals and negative half-integer
// powers through these): a user's h(x) may legitimately divide by something that is 0 or overflows to infinity in a degenerate
// configuration (the MSCKF test model does: a landmark scaled out to 1e200), where IEEE gives inf / 0 and the Newton steps turn the
// hardware seed's exact inf / 0 into NaN (inf * 0).  The seed IS the IEEE answer there, so it is returned whenever the refined value
// is not finite: one compare + select at the end of the chain.  The factorisations keep the unguarded forms (their pivots are variances).
__device__ __forceinline__ double safe_recip(const double d) {
#ifdef RN_EXACT_MATH
  return 1.0 / d;
#endif
  const double r0 = __builtin_amdgcn_rcp(d);
  double r = fma(fma(-d, r0, 1.0), r0, r0);
  r = fma(fma(-d, r, 1.0), r, r);
  return (r - r == 0.0) ? r : r0;
}
__device__ __forceinline__ double safe_rsqrt(const double a) {
#ifdef RN_EXACT_MATH
  return 1.0 / sqrt(a);
#endif
  const double y0 = __builtin_amdgcn_rsq(a);
  const double h = 0.5 * a;
  double y = fma(y0, fma(-h * y0, y0, 0.5), y0);
  y = fma(y, fma(-h * y, y, 0.5), y);
  return (y - y == 0.0) ? y : y0;
}

// a^(-N/2) for odd N (sympy's pow(r2, -1.5), pow(r2, -2.5) of gravity-like terms): powers of the reciprocal square root instead of an
// IEEE sqrt followed by an IEEE division -- two chains of ~20 and ~12 DEPENDENT fp64 instructions on the one lane per filter that
// evaluates the model's scalars (40 cycles each for a lone wavefront), against 7 + (N + 1) / 2 here.  Within an ulp or two of the
// correctly rounded composition (tests/test_gpu_parity.py::test_scalar_sympy_routines_on_gpu bounds it against the oracle's libm).
template <int N>
__device__ __forceinline__ double rsqrt_pow(const double a) {
  static_assert(N >= 1 && N <= 9 && (N & 1), "odd powers of 1 / sqrt(a)");
  const double r = safe_rsqrt(a);
  const double r2 = r * r;
  double p = r;
#pragma unroll
  for (int i = 0; i < (N - 1) / 2; i++) p *= r2;
  return p;
}

// ---- left null space of the extra-argument Jacobian (MSCKF, /root/reference/rednose/templates/ekf_c.c:66-76) --------
// The reference projects the residual, H and R of a feature-track observation on A = kernel(Hea^T) (Eigen fullPivLu;
// numpy twin: SVD null space, ekf_sym.py:20-26,583).  Any basis of that null space gives the same x and P; we use the
// orthonormal one that falls out of a Householder QR of Hea (Z x A): Hea = Q [R; 0], A_basis = Q[:, A:], so that
// "A^T v" is (Q^T v)[A:], i.e. A reflections applied to v and the first A entries dropped -- A is never formed.
// u holds the A reflector vectors (entries below the pivot row are meaningful, the rest zero), beta[c] = 2 / (u_c.u_c).
// Returns false when Hea is rank deficient (a pivot column vanishes against the largest one: the reference's numpy path
// ignores such a measurement, ekf_sym.py:589-591); the reflectors are then the identity (beta = 0).
template <int Z, int A>
__device__ __forceinline__ bool householder_qr(double (&M)[Z * A], double (&u)[A * Z], double (&beta)[A]) {
  bool full_rank = true;
  double scale = 0.0;
#pragma unroll
  for (int c = 0; c < A; c++) {
    double nrm2 = 0.0;
#pragma unroll
    for (int i = c; i < Z; i++) nrm2 += M[i * A + c] * M[i * A + c];
    const double nrm = sqrt(nrm2);
    if (c == 0) scale = nrm;
    if (!(nrm > 2.220446049250313e-16 * Z * scale) || !(scale > 0.0)) full_rank = false;
    const double alpha = M[c * A + c] > 0.0 ? -nrm : nrm;
#pragma unroll
    for (int i = 0; i < Z; i++) u[c * Z + i] = i < c ? 0.0 : M[i * A + c];
    u[c * Z + c] -= alpha;
    double uu = 0.0;
#pragma unroll
    for (int i = c; i < Z; i++) uu += u[c * Z + i] * u[c * Z + i];
    beta[c] = uu > 0.0 ? 2.0 / uu : 0.0;
#pragma unroll
    for (int j = c + 1; j < A; j++) {
      double dot = 0.0;
#pragma unroll
      for (int i = c; i < Z; i++) dot += u[c * Z + i] * M[i * A + j];
      dot *= beta[c];
#pragma unroll
      for (int i = c; i < Z; i++) M[i * A + j] -= dot * u[c * Z + i];
    }
  }
  if (!full_rank) {
#pragma unroll
    for (int c = 0; c < A; c++) beta[c] = 0.0;
  }
  return full_rank;
}

// v <- Q^T v for the reflectors above (u, beta may live in LDS: broadcast reads)
template <int Z, int A>
__device__ __forceinline__ void apply_reflectors(const double* u, const double* beta, double (&v)[Z]) {
#pragma unroll
  for (int c = 0; c < A; c++) {
    double dot = 0.0;
#pragma unroll
    for (int i = c; i < Z; i++) dot += u[c * Z + i] * v[i];
    dot *= beta[c];
#pragma unroll
    for (int i = c; i < Z; i++) v[i] -= dot * u[c * Z + i];
  }
}

// R <- Q^T R Q (Z x Z, row-major, in registers); the projected noise A^T R A is its trailing (Z - A) x (Z - A) block
template <int Z, int A>
__device__ __forceinline__ void project_noise(const double (&u)[A * Z], const double (&beta)[A], double (&R)[Z * Z]) {
  double v[Z];
#pragma unroll
  for (int j = 0; j < Z; j++) {          // columns
#pragma unroll
    for (int i = 0; i < Z; i++) v[i] = R[i * Z + j];
    apply_reflectors<Z, A>(u, beta, v);
#pragma unroll
    for (int i = 0; i < Z; i++) R[i * Z + j] = v[i];
  }
#pragma unroll
  for (int i = 0; i < Z; i++) {          // rows
#pragma unroll
    for (int j = 0; j < Z; j++) v[j] = R[i * Z + j];
    apply_reflectors<Z, A>(u, beta, v);
#pragma unroll
    for (int j = 0; j < Z; j++) R[i * Z + j] = v[j];
  }
}

// S = L D L^T with unit lower-triangular L (lower triangle of S is read); iD[j] = 1 / D[j].  Square-root-free on
// purpose: Cholesky's sqrt + reciprocal per column are two long dependent chains (see fast_recip).  S is the SPD
// innovation covariance; the reference solves with fullPivLu (ekf_c.c:89,101) -- same solution up to rounding.
template <int Z>
__device__ __forceinline__ void spd_factor(const double (&S)[Z * Z], double (&L)[Z * Z], double (&iD)[Z]) {
  double D[Z];
#pragma unroll
  for (int j = 0; j < Z; j++) {
    double d = S[j * Z + j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= L[j * Z + k] * L[j * Z + k] * D[k];
    D[j] = d;
    iD[j] = fast_recip(d);
#pragma unroll
    for (int i = j + 1; i < Z; i++) {
      double s = S[i * Z + j];
#pragma unroll
      for (int k = 0; k < j; k++) s -= L[i * Z + k] * L[j * Z + k] * D[k];
      L[i * Z + j] = s * iD[j];
    }
  }
}

// b <- L^{-1} b (unit lower triangular).  The quadratic form b^T S^{-1} b is then sum_i b[i]^2 iD[i].
template <int Z>
__device__ __forceinline__ void spd_forward(const double (&L)[Z * Z], const double (&iD)[Z], double (&b)[Z]) {
  (void)iD;
#pragma unroll
  for (int i = 0; i < Z; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= L[i * Z + k] * b[k];
    b[i] = s;
  }
}

// b <- (L D L^T)^{-1} b
template <int Z>
__device__ __forceinline__ void spd_solve(const double (&L)[Z * Z], const double (&iD)[Z], double (&b)[Z]) {
  spd_forward<Z>(L, iD, b);
#pragma unroll
  for (int i = Z - 1; i >= 0; i--) {
    double s = b[i] * iD[i];
#pragma unroll
    for (int k = i + 1; k < Z; k++) s -= L[k * Z + i] * b[k];
    b[i] = s;
  }
}

// General (non-symmetric) variant for the step-granular kernels: S = L D U with unit lower-triangular L (stored below the diagonal
// of LU) and unit upper-triangular U (above it), no pivoting; iD[j] = 1 / D[j].  The reference never symmetrises P
// (ekf_c.c:24,115), so its S = H P H^T + R (ekf_c.c:100) is as asymmetric as the caller's P and is solved as a general matrix
// (fullPivLu, ekf_c.c:101): reading one triangle of S, as spd_factor does, would change the result by the asymmetry of P.  For a
// symmetric S this IS the L D L^T above (U = L^T) at the same dependent-chain depth: the entries of U are formed next to those of L.
// No pivoting: S is a covariance plus R up to that asymmetry -- its symmetric part is positive definite.
template <int Z>
__device__ __forceinline__ void ldu_factor(const double (&S)[Z * Z], double (&LU)[Z * Z], double (&iD)[Z]) {
  double D[Z];
#pragma unroll
  for (int j = 0; j < Z; j++) {
    double d = S[j * Z + j];
#pragma unroll
    for (int k = 0; k < j; k++) d -= LU[j * Z + k] * D[k] * LU[k * Z + j];
    D[j] = d;
    iD[j] = fast_recip(d);
#pragma unroll
    for (int i = j + 1; i < Z; i++) {
      double s = S[i * Z + j], u = S[j * Z + i];
#pragma unroll
      for (int k = 0; k < j; k++) {
        s -= LU[i * Z + k] * D[k] * LU[k * Z + j];
        u -= LU[j * Z + k] * D[k] * LU[k * Z + i];
      }
      LU[i * Z + j] = s * iD[j];
      LU[j * Z + i] = u * iD[j];
    }
  }
}

// b <- L^{-1} b
template <int Z>
__device__ __forceinline__ void ldu_forward(const double (&LU)[Z * Z], const double (&iD)[Z], double (&b)[Z]) {
  (void)iD;
#pragma unroll
  for (int i = 0; i < Z; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= LU[i * Z + k] * b[k];
    b[i] = s;
  }
}

// b <- U^{-T} b.  The quadratic form b^T S^{-1} b = (U^{-T} b)^T D^{-1} (L^{-1} b) = sum_i u[i] v[i] iD[i].
template <int Z>
__device__ __forceinline__ void ldu_forward_t(const double (&LU)[Z * Z], const double (&iD)[Z], double (&b)[Z]) {
  (void)iD;
#pragma unroll
  for (int i = 0; i < Z; i++) {
    double s = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) s -= LU[k * Z + i] * b[k];
    b[i] = s;
  }
}

// b <- (L D U)^{-1} b
template <int Z>
__device__ __forceinline__ void ldu_solve(const double (&LU)[Z * Z], const double (&iD)[Z], double (&b)[Z]) {
  ldu_forward<Z>(LU, iD, b);
#pragma unroll
  for (int i = Z - 1; i >= 0; i--) {
    double s = b[i] * iD[i];
#pragma unroll
    for (int k = i + 1; k < Z; k++) s -= LU[i * Z + k] * b[k];
    b[i] = s;
  }
}

// EKFSym::normalize_slice (/root/reference/rednose/helpers/ekf_sym.cc:75-77): x[idx:idx+4] /= ||.||
template <int DIM>
__device__ __forceinline__ void normalize_quat(double (&x)[DIM], int idx) {
  // one reciprocal square root (v_rsq_f64 + two Newton steps, fast_rsqrt) and four multiplications: the IEEE sqrt + four IEEE
  // divisions this replaces were ~30 dependent fp64 instructions on the one lane per filter that runs it, twice per step (after
  // predict and after the update) -- about a microsecond of every step of the fused run.  Differs from Eigen's normalize() in the last bit.
  const double r = fast_rsqrt(x[idx] * x[idx] + x[idx + 1] * x[idx + 1] + x[idx + 2] * x[idx + 2] + x[idx + 3] * x[idx + 3]);
  x[idx] *= r;
  x[idx + 1] *= r;
  x[idx + 2] *= r;
  x[idx + 3] *= r;
}

// ------------------------------------------------------------------------------------------------
// per-filter checkpoint rings (orchestrators with per-filter timelines): `rec` doubles of record i of a flat array (row
// stride flat_stride) <-> entry slot[i] of filter i in a ring laid out (K, n, ring_stride), for the filters with
// active[i] != 0 (NULL: all).  One wavefront per filter and grid-stride; a record is contiguous on both sides.  The reference
// keeps one (x, P, observation) checkpoint list PER FILTER INSTANCE (/root/reference/rednose/helpers/ekf_sym.py:440-450,
// ekf_sym.cc:142-156); in a batch every filter may be at a different position of its own ring, hence the slot vector.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ring_copy(double* __restrict__ ring, const int64_t ring_stride, double* __restrict__ flat,
                                                  const int64_t flat_stride, const int64_t rec, const int32_t* __restrict__ slot,
                                                  const uint8_t* __restrict__ active, const int64_t n, const int to_ring) {
  for (int64_t f = blockIdx.x; f < n; f += gridDim.x) {
    if (active != nullptr && active[f] == 0) continue;
    double* r = ring + ((int64_t)slot[f] * n + f) * ring_stride;
    double* a = flat + f * flat_stride;
    if (to_ring) {
      for (int64_t i = threadIdx.x; i < rec; i += 64) r[i] = a[i];
    } else {
      for (int64_t i = threadIdx.x; i < rec; i += 64) a[i] = r[i];
    }
  }
}

// Packed lower triangles <-> full symmetric matrices, `count` records of E (E + 1) / 2 / E * E doubles (the packed-triangle trace of
// batch_run_tri / batch_rts_tri against the full-matrix interfaces).  One thread per entry of the full matrix.
template <int E>
__global__ void k_tri_unpack(const double* __restrict__ tri, double* __restrict__ full, const int64_t count) {
  constexpr int TRI = E * (E + 1) / 2;
  const int64_t total = count * E * E;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rec = i / (E * E);
    const int e = (int)(i - rec * (E * E)), r = e / E, c = e - r * E;
    const int hi = r > c ? r : c, lo = r > c ? c : r;
    full[i] = tri[rec * TRI + (hi * (hi + 1)) / 2 + lo];
  }
}
template <int E>
__global__ void k_tri_pack(const double* __restrict__ full, double* __restrict__ tri, const int64_t count) {
  constexpr int TRI = E * (E + 1) / 2;
  const int64_t total = count * TRI;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t rec = i / TRI;
    const int p = (int)(i - rec * TRI);
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= p) r++;
    tri[i] = full[rec * (E * E) + r * E + (p - (r * (r + 1)) / 2)];      // the LOWER triangle, like batch_rts reads it
  }
}

// flags[i] = value for the filters with mask[i] != 0 (the orchestrators' "observation too old for this filter's ring, ignored": bits 4 | 5 on top
// of what the launch wrote), on the stream, without a round trip of the flag bytes through the host
__global__ void k_flags_set(uint8_t* __restrict__ flags, const uint8_t* __restrict__ mask, const int value, const int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (mask[i] != 0) flags[i] = (uint8_t)value;
  }
}

// ------------------------------------------------------------------------------------------------
// host-side staging for the single-filter host-pointer entry points (the reference's scalar ABI)
// ------------------------------------------------------------------------------------------------
// One buffer of PINNED host memory mapped into the device's address space: an entry point packs its arguments into `host` with memcpy,
// launches the batch-of-one kernel on `dev` (the same bytes as the GPU addresses them: the kernel reads and writes host memory in place
// over PCIe), waits for the null stream and unpacks.  No staging copies: seven blocking hipMemcpy calls of a few hundred bytes each made a
// predict + update of ONE filter 193 us (81 + 112), whatever the model; profiles/r6_scalar_abi_latency.txt.
struct Scratch {
  std::mutex mu;
  double* host = nullptr;
  double* dev = nullptr;
  size_t cap = 0;   // doubles
  int ensure(size_t doubles) {
    if (doubles <= cap) return OK;
    if (host) (void)hipHostFree(host);
    host = dev = nullptr;
    cap = 0;
    const size_t want = doubles < 4096 ? 4096 : doubles;
    RN_HIP(hipHostMalloc(reinterpret_cast<void**>(&host), want * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    RN_HIP(hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), host, 0));
    cap = want;
    return OK;
  }
  // the launches of the entry points go to the null stream; their results are in `host` once it has drained
  int wait(const char* what, int line) {
    const hipError_t e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { fail(ERR_HIP, (int)e, what, line); return ERR_HIP; }
    return OK;
  }
  void put(size_t off, const double* src, size_t n) { std::memcpy(host + off, src, n * sizeof(double)); }
  void get(double* dst, size_t off, size_t n) const { std::memcpy(dst, host + off, n * sizeof(double)); }
};

inline Scratch& scratch() {
  static Scratch s;
  return s;
}

}  // namespace rn
