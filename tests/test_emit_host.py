"""The arithmetic the lane-per-filter emitter generates (emit_small.predict_regs / update_*_regs: the device functions every kernel of
that family calls -- step kernels, fused runs, the gate) compiled FOR THE HOST and checked against the oracle, without a GPU.

The generated functions are plain C++ over register arrays (all sparsity resolved at generation time) plus three templates of
the runtime header (LDL^T factorisation and substitutions of the Z x Z innovation covariance).  Here their text is wrapped in a
translation unit in which `__device__` is empty and the device reciprocal (v_rcp_f64 + two Newton steps, ~1 ulp) is 1.0 / d, built
with g++ and driven through ctypes on seeded random filters; the oracle (reference-generated sympy C + C restatement of ekf_c.c)
runs the same fused predict + update.  What this pins on every CPU run: the emitter's algebra (F P F^T + dt Q with structural
zeros dropped, the rank-Z Joseph form, the gate's R *= 1e16 path and its flag), for linear, nonlinear and affine models."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import assert_close

HDR = os.path.join(os.path.dirname(__file__), "..", "rednose_amd", "templates", "ekf_hip_rt.h")
pytestmark = pytest.mark.timeout(180, method="thread")      # the lane emulations below wait on barriers: a mismatch must fail, not hang

# The lanes of an emulated wavefront (workgroup) are FIBERS of one OS thread, not threads: a barrier between 64 threads on a handful of cores is
# a round of futex sleeps (~0.2 ms; the smoother's emulation executes ~6 000 of them per step), between fibers it is 64 stack switches.  The
# translation units keep the pthread spelling; this text replaces <pthread.h>: pthread_create registers a fiber, the first pthread_join runs them
# all, pthread_barrier_wait blocks the caller until its barrier is complete, sched_yield hands over.  WHICH runnable fiber continues is drawn
# from a seeded generator at every hand-over, so between two barriers the lanes run whole, in an arbitrary order: a read that is not separated by
# a barrier from another lane's write (or a write from another lane's read) sees the wrong value for about half of the lane pairs -- sharper than
# preemptive threads running in near-lockstep, and the same on every run.
_FIBERS = r"""
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <sys/mman.h>
extern "C" void rn_fib_switch(void** save_sp, void* load_sp) __attribute__((visibility("hidden")));
asm(R"ASM(
.text
.hidden rn_fib_switch
.globl rn_fib_switch
.type rn_fib_switch,@function
rn_fib_switch:
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
.size rn_fib_switch,.-rn_fib_switch
)ASM");
namespace fib {
constexpr int MAXF = 256, LOCAL = 128;
constexpr size_t STACK = size_t(2) << 20;
struct Fiber { void* sp; void* (*fn)(void*); void* arg; bool done, blocked; char local[LOCAL]; };
struct Local { void* p; int n, off; };
static Fiber g_f[MAXF + 1];                    // [MAXF]: the caller's context
static Local g_loc[16];
static int g_nloc = 0, g_locbytes = 0, g_n = 0, g_cur = MAXF, g_alive = 0;
static char* g_stacks = nullptr;
static uint32_t g_rng = 2463534242u;
struct RegisterLocal { RegisterLocal(void* p, int n) { g_loc[g_nloc++] = Local{p, n, g_locbytes}; g_locbytes += (n + 7) & ~7; if (g_locbytes > LOCAL || g_nloc > 16) abort(); } };
static void to(int nx) {                       // fiber-local variables travel with their fiber
  const int me = g_cur;
  if (nx == me) return;
  for (int i = 0; i < g_nloc; i++) { std::memcpy(g_f[me].local + g_loc[i].off, g_loc[i].p, g_loc[i].n); std::memcpy(g_loc[i].p, g_f[nx].local + g_loc[i].off, g_loc[i].n); }
  g_cur = nx;
  rn_fib_switch(&g_f[me].sp, g_f[nx].sp);
}
static int pick() {
  int runnable = 0;
  for (int i = 0; i < g_n; i++) runnable += !g_f[i].done && !g_f[i].blocked;
  if (!runnable) { std::fprintf(stderr, "host emulation: every lane waits at a barrier that cannot complete\n"); abort(); }
  g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5;
  int k = (int)(g_rng % (uint32_t)runnable);
  for (int i = 0; i < g_n; i++) if (!g_f[i].done && !g_f[i].blocked && k-- == 0) return i;
  return -1;
}
static void yield() { to(pick()); }
static void entry() {
  Fiber& me = g_f[g_cur];
  me.fn(me.arg);
  me.done = true;
  to(--g_alive == 0 ? MAXF : pick());
  abort();
}
static void spawn(void* (*fn)(void*), void* arg) {
  if (!g_stacks) g_stacks = static_cast<char*>(mmap(nullptr, STACK * MAXF, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
  if (g_stacks == MAP_FAILED || g_n >= MAXF) abort();
  Fiber& f = g_f[g_n];
  f.fn = fn; f.arg = arg; f.done = f.blocked = false;
  std::memset(f.local, 0, LOCAL);
  void** top = reinterpret_cast<void**>(g_stacks + STACK * (g_n + 1));      // 16-byte aligned; [-1]: where a return address would be, [-2]: entry, [-8 .. -3]: the six saved registers
  top[-1] = nullptr; top[-2] = reinterpret_cast<void*>(&entry);
  for (int i = 3; i <= 8; i++) top[-i] = nullptr;
  f.sp = top - 8;
  g_n++; g_alive++;
}
static void run_all() {
  if (g_alive) to(pick());
  g_n = 0;
}
struct Barrier { int need, count, waiting[MAXF]; };
static void wait(Barrier* b) {
  if (b->count + 1 == b->need) { for (int i = 0; i < b->count; i++) g_f[b->waiting[i]].blocked = false; b->count = 0; return; }
  b->waiting[b->count++] = g_cur;
  g_f[g_cur].blocked = true;
  yield();
}
}  // namespace fib
inline int fib_barrier_init(fib::Barrier* b, const void*, int n) { b->need = n; b->count = 0; return 0; }
inline int fib_barrier_wait(fib::Barrier* b) { fib::wait(b); return 0; }
inline int fib_barrier_destroy(fib::Barrier*) { return 0; }
inline int fib_create(int* t, const void*, void* (*fn)(void*), void* arg) { *t = fib::g_n; fib::spawn(fn, arg); return 0; }
inline int fib_join(int, void**) { fib::run_all(); return 0; }
inline int fib_yield() { fib::yield(); return 0; }
#define pthread_barrier_t fib::Barrier
#define pthread_t int
#define pthread_barrier_init fib_barrier_init
#define pthread_barrier_wait fib_barrier_wait
#define pthread_barrier_destroy fib_barrier_destroy
#define pthread_create fib_create
#define pthread_join fib_join
#define sched_yield fib_yield
"""



def _once(build):
  """A host library is built once per (builder, model, variant) and session: parametrisations of one test share the loaded object."""
  import functools
  libs = {}

  @functools.wraps(build)
  def cached(tmp_path, spec, *a, **kw):
    key = (spec.name, spec.dim_x, spec.dim_err, tuple((k.kind, k.zdim, k.maha_test, k.maha_thresh) for k in spec.kinds), str(spec.f_sym), a, tuple(sorted(kw.items())))
    if key not in libs:
      libs[key] = build(tmp_path, spec, *a, **kw)
    return libs[key]
  return cached

def _fiberize(src):
  """The host translation unit with its lanes as fibers (_FIBERS): <pthread.h> / <sched.h> replaced, `thread_local` variables made fiber-local."""
  assert "#include <pthread.h>" in src
  src = src.replace("#include <pthread.h>", _FIBERS, 1).replace("#include <sched.h>", "")

  def local(m):
    names = [n.strip() for n in m.group(2).split(",")]
    init = m.group(3) or ""
    return f"static {m.group(1)} {', '.join(n + init for n in names)};" + "".join(f" static fib::RegisterLocal fib_local_{n}(&{n}, sizeof({n}));" for n in names)
  src = re.sub(r"static thread_local (\w+) ([\w, ]+?)( = \w+)?;", local, src)
  assert "thread_local" not in src
  return src


def _function_text(text, name):
  """Text of the function `name` of the runtime header, from its `template <...>` line (when it has one) to its closing brace."""
  if name == "sincos_fast":      # not in the runtime header: emitted into the generated file of a model with trigonometric terms
    from rednose_amd.codegen.lower import SINCOS_FAST
    text = SINCOS_FAST
  at = text.index(f" {name}(")
  start = text.rfind("\n", 0, at) + 1
  prev = text.rfind("\n", 0, start - 1) + 1
  if text[prev:start].startswith("template <"):
    start = prev
  depth, i = 0, text.index("{", at)
  while True:
    depth += {"{": 1, "}": -1}.get(text[i], 0)
    i += 1
    if depth == 0:
      return text[start:i]


@_once
def _host_library(tmp_path, spec, sym=False):
  """sym=True: the `_sym` flavour the fused multi-step kernels call, behind the symmetrisation they apply when the state enters the
  registers (emit_small.predict_regs)."""
  from rednose_amd.codegen import emit_small
  hdr = open(HDR, encoding="utf-8").read()
  helpers = "\n".join(_function_text(hdr, f) for f in ("spd_factor", "spd_forward", "spd_solve", "ldu_factor", "ldu_forward", "ldu_forward_t", "ldu_solve", "rsqrt_pow", "sincos_fast", "normalize_quat"))
  D, E = spec.dim_x, spec.dim_err
  body = [emit_small.predict_regs(spec, sym)[0]] + [emit_small.update_regs(spec, k, sym)[0] for k in spec.kinds]
  sfx = "_sym" if sym else ""
  symm = (f"  for (int i = 0; i < {E}; i++) for (int j = i + 1; j < {E}; j++) {{ P[i * {E} + j] = 0.5 * (P[i * {E} + j] + P[j * {E} + i]); "
          f"P[j * {E} + i] = P[i * {E} + j]; }}") if sym else ""
  quat = "".join(f" rn::normalize_quat<{D}>(x, {q});" for q in spec.quaternion_idxs)
  entry = []
  for k in spec.kinds:
    Z = k.zdim
    assert k.ea_sym is None, "kinds with extra arguments are lane-group (MSCKF) kinds"
    entry.append(f"""
extern "C" int host_step_{k.kind}(double* gx, double* gP, const double* Q, double dt, double* gz, const double* gR) {{
  double x[{D}], P[{E * E}], z[{Z}], R[{Z * Z}];
  for (int i = 0; i < {D}; i++) x[i] = gx[i];
  for (int i = 0; i < {E * E}; i++) P[i] = gP[i];
  for (int i = 0; i < {Z}; i++) z[i] = gz[i];
  for (int i = 0; i < {Z * Z}; i++) R[i] = gR[i];
{symm}
  predict_regs{sfx}(x, P, Q, dt);{quat}
  const int fl = update_{k.kind}_regs{sfx}(x, P, z, R);{quat}
  for (int i = 0; i < {D}; i++) gx[i] = x[i];
  for (int i = 0; i < {E * E}; i++) gP[i] = P[i];
  for (int i = 0; i < {Z}; i++) gz[i] = z[i];
  return fl;
}}""")
  src = "\n".join(["#include <cmath>", "#include <cstdint>", "#define __device__", "#define __forceinline__ inline",
                   "namespace rn {", "inline double fast_recip(const double d) { return 1.0 / d; }      // device: v_rcp_f64 + two Newton steps",
                   "inline double fast_rsqrt(const double a) { return 1.0 / std::sqrt(a); }      // device: v_rsq_f64 + two Newton steps",
                   "inline double safe_recip(const double d) { return 1.0 / d; }", "inline double safe_rsqrt(const double a) { return 1.0 / std::sqrt(a); }",
                   helpers, "}  // namespace rn"] + body + entry)
  cpp, lib = tmp_path / f"{spec.name}{sfx}_host.cpp", tmp_path / f"lib{spec.name}{sfx}_host.so"
  cpp.write_text(_fiberize(src) if "#include <pthread.h>" in src else src, encoding="utf-8")
  res = subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", str(cpp), "-o", str(lib)], capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-3000:]
  return ctypes.CDLL(str(lib))


def _model(name):
  if name == "kinematic":
    from examples.kinematic_kf import KinematicKalman as M
    return M, M.model(), {}
  if name in ("kinematic6", "kinematic6_maha"):
    from examples.kinematic6_kf import Kinematic6Kalman as M
    mdl = M.model()
    mdl["name"] = name
    return M, mdl, ({"maha_test_kinds": [1]} if name.endswith("maha") else {})
  import examples.random_kf as R
  M = getattr(R, {"rand3": "Random3Kalman", "rand5": "Random5Kalman", "randaff5": "RandomAffine5Kalman"}[name])
  return M, M.model(), {}


@pytest.mark.parametrize("sym", [False, True], ids=["full", "sym"])
@pytest.mark.parametrize("name", ["kinematic", "kinematic6", "kinematic6_maha", "rand3", "rand5", "randaff5"])
def test_generated_lane_per_filter_arithmetic_on_the_host(tmp_path, name, sym):
  """Both flavours of the generated functions against the oracle, on symmetric AND on deliberately asymmetric covariances
  (SPD + a skew part of 1e-3 of its scale).  The reference's products use both halves of P and never symmetrise
  (ekf_c.c:24,101,115): the full flavour (step-granular kernels) must follow it entry for entry on the asymmetric input; the `_sym`
  flavour (fused multi-step kernels) is specified on (P + P^T) / 2 and must match the oracle run on THAT matrix."""
  from oracle_lib import OracleLib
  from rednose_amd.codegen.spec import build_spec
  M, mdl, kw = _model(name)
  spec = build_spec(**mdl, **kw)
  assert re.fullmatch(r"[a-z0-9_]+", spec.name)
  lib = _host_library(tmp_path, spec, sym)
  o = OracleLib(name)
  D, E = spec.dim_x, spec.dim_err
  rng = np.random.default_rng(len(name) + D)
  n = 40
  x_init = np.asarray(getattr(M, "initial_x", np.zeros(D)), dtype=np.float64)
  P_init = np.diag(getattr(M, "initial_P_diag", np.ones(E)))
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  dp = ctypes.POINTER(ctypes.c_double)
  gated = 0
  for k in spec.kinds:
    Z = k.zdim
    R = np.ascontiguousarray(np.atleast_2d(M.obs_noise[k.kind]), dtype=np.float64)
    x0 = x_init[None] + rng.normal(size=(n, D)) * 0.3
    A = rng.normal(size=(n, E, E)) * 0.2
    P0s = P_init[None] + A @ A.transpose(0, 2, 1)
    W = rng.normal(size=(n, E, E))
    P0a = P0s + 1e-3 * np.abs(P0s).max(axis=(1, 2), keepdims=True) * (W - W.transpose(0, 2, 1))
    # a few far-off observations: with a gated kind some filters take the R *= 1e16 path (flag 1), the others do not
    z0 = rng.normal(size=(n, Z)) * np.where(rng.uniform(size=(n, 1)) < 0.3, 40.0, 0.5)
    for dt in (0.0, 0.02):
      for P0, tag in ((P0s, "symmetric P"), (P0a, "asymmetric P")):
        Pin = 0.5 * (P0 + P0.transpose(0, 2, 1)) if sym else P0        # what the flavour is specified on
        xr, Pr, zr = x0.copy(), Pin.copy(), z0.copy()
        fr = np.zeros(n, dtype=np.uint8)
        o.batch_step(k.kind, xr, Pr, zr, R, Q, dt, flags=fr)
        xh, Ph, zh = x0.copy(), P0.copy(), z0.copy()
        fh = np.zeros(n, dtype=np.uint8)
        fn = getattr(lib, f"host_step_{k.kind}")
        fn.argtypes = [dp, dp, dp, ctypes.c_double, dp, dp]
        for i in range(n):
          fh[i] = fn(xh[i].ctypes.data_as(dp), Ph[i].ctypes.data_as(dp), Q.ctypes.data_as(dp), dt, zh[i].ctypes.data_as(dp), R.ctypes.data_as(dp))
        what = f"{name} kind {k.kind} dt {dt} {tag} ({'sym' if sym else 'full'})"
        assert np.array_equal(fh & 1, fr & 1), what + " gate flags"
        gated += int((fh & 1).sum())
        assert_close(xh, xr, rtol=1e-11, floor=1e-13, what=what + " x")
        assert_close(Ph.reshape(n, -1), Pr.reshape(n, -1), rtol=1e-11, floor=1e-13, what=what + " P")
        assert_close(zh, zr, rtol=1e-11, atol=1e-13 * max(1.0, np.abs(z0).max()), what=what + " y")
        if sym:
          assert np.array_equal(Ph, Ph.transpose(0, 2, 1)), what + ": result not exactly symmetric"
  assert (gated > 0) == name.endswith("maha")


# ---- lane-group family: the three phases of the step kernels (emit_wide2) emulated lane by lane ---------------------------------
# Phase 1 / 3 functions (scal_*) are scalar code per filter.  The matrix-phase functions (mat_predict, mat_update_*) are written
# for the lanes of one filter's group working on ONE image of P in LDS, ordered only by rn::wave_lds_sync(): no cross-lane
# instruction at all.  On the host every lane is a thread and wave_lds_sync() a barrier over the group, the LDS arrays are plain
# shared arrays -- the text that runs is the text the kernels inline, under the model's own tuning defaults (live: the
# register-lean structure with rows of P in LDS).


@_once
def _wide_host_library(tmp_path, spec):
  from rednose_amd.codegen import emit_wide2, tuning
  hdr = open(HDR, encoding="utf-8").read()
  helpers = "\n".join(_function_text(hdr, f) for f in ("spd_factor", "spd_forward", "spd_solve", "ldu_factor", "ldu_forward", "ldu_forward_t", "ldu_solve", "rsqrt_pow", "sincos_fast", "normalize_quat"))
  with tuning.using_model(spec):
    text, lay = emit_wide2.device_functions(spec)
    GL = emit_wide2.group_lanes(spec)
  D, E = spec.dim_x, spec.dim_err
  kinds = [k for k in spec.kinds if k.He_sym is None and k.ea_sym is None]
  want = {"scal_predict", "scal_keep", "scal_inject", "mat_predict"} | {f"scal_obs_{k.kind}" for k in kinds} | {f"mat_update_{k.kind}" for k in kinds}
  fns = []
  for fn in re.split(r"\n(?=(?:template <[^\n]*>\n)?__device__ )", text):
    m = re.search(r"__device__ \w+ \w+ (\w+)\(", fn)
    if m and m.group(1) in want:
      fns.append(fn)
      want.discard(m.group(1))
  assert not want, want
  predict_sig = next(f for f in fns if " mat_predict(" in f).split("{")[0]
  q_arg = "Q" if "const double* sQ" in predict_sig else "qcol"
  entry = []
  for k in kinds:
    Z = k.zdim
    entry.append(f"""
struct Job{k.kind} {{ double* sP; const double* Q; const double* R; double* sl; double* sG; double* sK; int c; int do_pred; }};
static void* lane_{k.kind}(void* p) {{
  const Job{k.kind}& j = *static_cast<Job{k.kind}*>(p);
  const bool act = j.c < {E};
  const int cc = act ? j.c : 0;
  double qcol[{E}];
  for (int i = 0; i < {E}; i++) qcol[i] = j.Q[i * {E} + cc];
  const double* Q = j.Q; (void)Q; (void)qcol;
  if (j.do_pred) mat_predict(j.sP, {q_arg}, j.sl, cc, act);
  mat_update_{k.kind}(j.sP, j.R, j.sl, j.sl, j.sG, j.sK, cc, act);
  return nullptr;
}}
extern "C" int host_wide_step_{k.kind}(double* x, double* P, const double* Q, double dt, double* z, const double* R, int norm_quats, int do_pred) {{
  static double sl[{lay.SLOT + 8}], sP[{E * E + 2}], sG[{Z * E + 2}], sK[{Z * E + 2}];
  g_sync_on = false;
  if (do_pred) scal_predict(x, dt, sl, norm_quats); else scal_keep(x, sl, norm_quats);
  scal_obs_{k.kind}(sl, z);
  for (int i = 0; i < {E * E}; i++) sP[i] = P[i];
  pthread_barrier_init(&g_bar, nullptr, {GL});
  g_sync_on = true;
  pthread_t th[{GL}];
  Job{k.kind} jobs[{GL}];
  for (int c = 0; c < {GL}; c++) {{ jobs[c] = Job{k.kind}{{sP, Q, R, sl, sG, sK, c, do_pred}}; pthread_create(&th[c], nullptr, lane_{k.kind}, &jobs[c]); }}
  for (int c = 0; c < {GL}; c++) pthread_join(th[c], nullptr);
  pthread_barrier_destroy(&g_bar);
  g_sync_on = false;
  const int fl = scal_inject(sl, x, norm_quats) | (int)sl[{lay.OFF_FL}];
  for (int i = 0; i < {Z}; i++) z[i] = sl[{lay.OFF_Y} + i];
  for (int i = 0; i < {E * E}; i++) P[i] = sP[i];
  return fl;
}}""")
  src = "\n".join(["#include <cmath>", "#include <cstdint>", "#include <pthread.h>", "#define __device__", "#define __forceinline__ inline",
                   "#define __noinline__", "#define __builtin_amdgcn_sched_barrier(x)", "static pthread_barrier_t g_bar;", "static bool g_sync_on = false;", "namespace rn {",
                   "inline void wave_lds_sync() { if (g_sync_on) pthread_barrier_wait(&g_bar); }      // device: a compiler fence inside one wavefront",
                   "inline double fast_recip(const double d) { return 1.0 / d; }", "inline double fast_rsqrt(const double a) { return 1.0 / std::sqrt(a); }", "inline double safe_recip(const double d) { return 1.0 / d; }", "inline double safe_rsqrt(const double a) { return 1.0 / std::sqrt(a); }", helpers, "}  // namespace rn"] + fns + entry)
  cpp, lib = tmp_path / f"{spec.name}_wide_host.cpp", tmp_path / f"lib{spec.name}_wide_host.so"
  cpp.write_text(_fiberize(src) if "#include <pthread.h>" in src else src, encoding="utf-8")
  res = subprocess.run(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", str(cpp), "-o", str(lib)], capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-3000:]
  return ctypes.CDLL(str(lib)), kinds


def _wide_model(name):
  if name in ("live", "live_maha"):
    from examples.live_kf import LiveKalman as M, ObservationKind as LK
    mdl = M.model() if hasattr(M, "model") else None
    return M, mdl, ({"maha_test_kinds": [LK.ECEF_POS]} if name == "live_maha" else {}), 3
  if name == "kinematic9":
    from examples.kinematic9_kf import Kinematic9Kalman as M
    return M, M.model(), {}, -1
  import examples.random_kf as R
  M = getattr(R, f"Random{name[4:]}Kalman")
  return M, M.model(), {}, -1


@pytest.mark.parametrize("name", ["kinematic9", "rand11", "rand24", "live", "live_maha"])
def test_generated_lane_group_step_on_the_host(tmp_path, name):
  from oracle_lib import OracleLib
  from rednose_amd.codegen.spec import build_spec
  M, mdl, kw, quat_idx = _wide_model(name)
  if mdl is None:
    pytest.skip("model definition not exposed as a dict")
  mdl = dict(mdl)
  mdl["name"] = name
  spec = build_spec(**mdl, **kw)
  lib, kinds = _wide_host_library(tmp_path, spec)
  o = OracleLib(name)
  D, E = spec.dim_x, spec.dim_err
  rng = np.random.default_rng(E)
  n = 6
  x_init = np.asarray(M.initial_x, dtype=np.float64)
  P_init = np.diag(M.initial_P_diag)
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  dp = ctypes.POINTER(ctypes.c_double)
  gated = 0
  for k in kinds:
    Z = k.zdim
    R = np.ascontiguousarray(np.atleast_2d(M.obs_noise.get(k.kind, 0.01 * np.eye(Z))), dtype=np.float64)      # (live defines no noise for its camera kinds)
    x0 = np.tile(x_init, (n, 1))
    x0 += rng.normal(size=(n, D)) * 0.01 * np.maximum(1.0, np.abs(x_init))[None] * (np.abs(x_init)[None] < 10.0)      # not the ECEF position: metres
    A = rng.normal(size=(n, E, E)) * 0.1 * np.sqrt(np.diag(P_init))[None, :, None]
    P0 = P_init[None] + A @ A.transpose(0, 2, 1)
    # observations near their prediction (h(x0) through the oracle), a third of them far off so that a gated kind splits
    hx = np.zeros((n, Z))
    for i in range(n):
      xq = x0[i].copy()
      if quat_idx >= 0:
        xq[quat_idx:quat_idx + 4] /= np.linalg.norm(xq[quat_idx:quat_idx + 4])
      o.call(f"h_{k.kind}", xq, np.zeros(4), hx[i])
    sig = np.sqrt(np.diag(R))[None]
    far = rng.uniform(size=(n, 1)) < 0.34        # (far against the PRIOR spread too: the position prior of live is 10 km wide)
    z0 = hx + rng.normal(size=(n, Z)) * sig + far * rng.normal(size=(n, Z)) * 40.0 * np.sqrt(P_init.max())
    # the step-granular kernels use both halves of P, like the reference (ekf_c.c:24,101,115): an asymmetric covariance -- skew part
    # 1e-3 of sqrt(P_ii P_jj) -- has to follow the oracle entry for entry as well
    Wk = rng.normal(size=(n, E, E))
    dg = np.sqrt(np.einsum("nii->ni", P0))
    P0a = P0 + 1e-3 * dg[:, :, None] * dg[:, None, :] * (Wk - Wk.transpose(0, 2, 1))
    for dt, Pin, tag in ((0.0, P0, ""), (0.01, P0, ""), (0.01, P0a, " asymmetric P"), (0.0, P0a, " asymmetric P")):
      xr, Pr, zr = x0.copy(), Pin.copy(), z0.copy()
      fr = np.zeros(n, dtype=np.uint8)
      o.batch_step(k.kind, xr, Pr, zr, R, Q, dt, quat_idx=quat_idx, flags=fr)
      xh, Ph, zh = x0.copy(), Pin.copy(), z0.copy()
      fh = np.zeros(n, dtype=np.uint8)
      fn = getattr(lib, f"host_wide_step_{k.kind}")
      fn.argtypes = [dp, dp, dp, ctypes.c_double, dp, dp, ctypes.c_int, ctypes.c_int]
      do_pred = int(not (dt == 0.0 and spec.identity_at_dt0()))
      for i in range(n):
        fh[i] = fn(xh[i].ctypes.data_as(dp), Ph[i].ctypes.data_as(dp), Q.ctypes.data_as(dp), dt, zh[i].ctypes.data_as(dp), R.ctypes.data_as(dp),
                   int(quat_idx >= 0), do_pred)
      what = f"{name} kind {k.kind} dt {dt}{tag}"
      assert np.array_equal(fh & 1, fr & 1), what + " gate flags"
      gated += int((fh & 1).sum())
      assert_close(xh, xr, rtol=1e-10, floor=1e-12, what=what + " x")
      assert_close(Ph.reshape(n, -1), Pr.reshape(n, -1), rtol=1e-9, floor=1e-11, what=what + " P")
      assert_close(zh, zr, rtol=1e-10, atol=1e-12 * max(1.0, np.abs(z0).max()), what=what + " y")
  assert (gated > 0) == (name == "live_maha")


# ---- lane-group fused run (emit_wide3): GL lanes x R rows per filter, rows of P in registers for T steps ----------------------------
# Same emulation (a thread per lane of one filter's group, wave_lds_sync() = barrier); the rows stay in each lane's "registers"
# (thread-local arrays) from step to step like in k_run, only x / P / z move through the shared images.  Both predict variants
# (general Q read through a pointer, diagonal Q in registers) run the same schedule.


@_once
def _run_host_library(tmp_path, spec):
  from rednose_amd.codegen import emit_wide2 as w2, emit_wide3 as w3, tuning
  hdr = open(HDR, encoding="utf-8").read()
  helpers = "\n".join(_function_text(hdr, f) for f in ("spd_factor", "spd_forward", "spd_solve", "ldu_factor", "ldu_forward", "ldu_forward_t", "ldu_solve", "rsqrt_pow", "sincos_fast", "normalize_quat"))
  D, E = spec.dim_x, spec.dim_err
  kinds = [k for k in spec.kinds if k.He_sym is None and k.ea_sym is None]
  with tuning.using_model(spec):
    GL, R, _ = w3.layout(spec)
    scal_text, lay = w2.device_functions(spec, lay_cls=w3.RunLayout, sfx="_r")
    fns = [scal_text, w3.predict_fn(spec), w3.predict_fn(spec, qdiag=True)] + [w3.update_fn(spec, k) for k in kinds]
  zmax = max(k.zdim for k in spec.kinds)
  rows = ", ".join(f"row{s}" for s in range(R))
  idx = ", ".join(f"rr{s}, rc{s}, ok{s}" for s in range(R))
  decl = "\n".join(f"  const int rr{s} = j.c + {GL * s}; const bool ok{s} = rr{s} < {E}; const int rc{s} = ok{s} ? rr{s} : 0;\n"
                   f"  double row{s}[{E}]; for (int q = 0; q < {E}; q++) row{s}[q] = j.sP[rc{s} * {E} + q];" for s in range(R))
  back = "\n".join(f"    if (ok{s}) for (int q = 0; q < {E}; q++) j.sP[rr{s} * {E} + q] = row{s}[q];" for s in range(R))
  qd = ", ".join(f"j.Q[rc{s} * {E + 1}]" for s in range(R))
  scal_cases = "\n".join(f"        case {k.kind}: scal_obs_{k.kind}_r(j.sl, j.sl + {lay.OFF_Y}); break;" for k in kinds)
  mat_cases = "\n".join(f"      case {k.kind}: update_{k.kind}_rows({rows}, j.R + t * {zmax * zmax}, j.sP, j.sG, j.sl, j.sl, {idx}); break;" for k in kinds)
  src = "\n".join(["#include <cmath>", "#include <cstdint>", "#include <pthread.h>", "#define __device__", "#define __forceinline__ inline",
                   "#define __noinline__", "static pthread_barrier_t g_bar;",
                   "static thread_local bool t_scalar = false;      // inside a scalar-phase function (one lane): its fences are not barriers",
                   "namespace rn {", "inline void wave_lds_sync() { if (!t_scalar) pthread_barrier_wait(&g_bar); }", "inline void pin(double&) {}",
                   "inline double fast_recip(const double d) { return 1.0 / d; }", "inline double fast_rsqrt(const double a) { return 1.0 / std::sqrt(a); }", "inline double safe_recip(const double d) { return 1.0 / d; }", "inline double safe_rsqrt(const double a) { return 1.0 / std::sqrt(a); }", helpers, "}  // namespace rn"] + fns + [f"""
struct Job {{ double* x; double* sP; const double* Q; const double* R; const int* kinds; const double* dts; double* z; int T; double* sl; double* sG;
             unsigned char* flags; int c; int norm_quats; int qdiag; int skip_dt0; }};
static void* lane(void* p) {{
  const Job& j = *static_cast<Job*>(p);
{decl}
  for (int t = 0; t < j.T; t++) {{
    const double dt = j.dts[t];
    const bool do_pred = !(j.skip_dt0 && dt == 0.0);
    if (j.c == 0) {{
      t_scalar = true;
      for (int i = 0; i < {zmax}; i++) j.sl[{lay.OFF_Y} + i] = j.z[t * {zmax} + i];
      if (do_pred) scal_predict_r(j.sl + {lay.OFF_X}, dt, j.sl, j.norm_quats); else scal_keep_r(j.sl + {lay.OFF_X}, j.sl, j.norm_quats);
      t_scalar = false;
    }}
    rn::wave_lds_sync();
    if (do_pred) {{
      if (j.qdiag) predict_rows_qd({rows}, j.sP, {qd}, j.sl, {idx});
      else predict_rows({rows}, j.sP, j.Q, j.sl, {idx});
    }}
    if (j.c == 0) {{
      t_scalar = true;
      switch (j.kinds[t]) {{
{scal_cases}
        default: break;
      }}
      t_scalar = false;
    }}
    rn::wave_lds_sync();
    switch (j.kinds[t]) {{
{mat_cases}
      default: break;
    }}
    if (j.c == 0) {{
      t_scalar = true;
      j.flags[t] = (unsigned char)(scal_inject_r(j.sl, j.sl + {lay.OFF_X}, j.norm_quats) | (int)j.sl[{lay.OFF_FL}]);
      for (int i = 0; i < {zmax}; i++) j.z[t * {zmax} + i] = j.sl[{lay.OFF_Y} + i];
      t_scalar = false;
    }}
    rn::wave_lds_sync();
  }}
{back}
  return nullptr;
}}
extern "C" void host_run(double* x, double* P, const double* Q, const double* R, const int* kinds, const double* dts, double* z, int T,
                         unsigned char* flags, int norm_quats, int qdiag, int skip_dt0) {{
  static double sl[{lay.SLOT + 8}], sP[{E * E + 2}], sG[{zmax * E + 8}];
  for (int i = 0; i < {D}; i++) sl[{lay.OFF_X} + i] = x[i];
  for (int i = 0; i < {E * E}; i++) sP[i] = P[i];
  pthread_barrier_init(&g_bar, nullptr, {GL});
  pthread_t th[{GL}];
  Job jobs[{GL}];
  for (int c = 0; c < {GL}; c++) {{ jobs[c] = Job{{x, sP, Q, R, kinds, dts, z, T, sl, sG, flags, c, norm_quats, qdiag, skip_dt0}}; pthread_create(&th[c], nullptr, lane, &jobs[c]); }}
  for (int c = 0; c < {GL}; c++) pthread_join(th[c], nullptr);
  pthread_barrier_destroy(&g_bar);
  for (int i = 0; i < {D}; i++) x[i] = sl[{lay.OFF_X} + i];
  for (int i = 0; i < {E * E}; i++) P[i] = sP[i];
}}"""])
  cpp, lib = tmp_path / f"{spec.name}_run_host.cpp", tmp_path / f"lib{spec.name}_run_host.so"
  cpp.write_text(_fiberize(src) if "#include <pthread.h>" in src else src, encoding="utf-8")
  res = subprocess.run(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", str(cpp), "-o", str(lib)], capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-3000:]
  return ctypes.CDLL(str(lib)), kinds, zmax


@pytest.mark.parametrize("name,qdiag", [("kinematic9", 1), ("rand11", 0), ("rand24", 1), ("live", 1), ("live", 0)])
def test_generated_lane_group_fused_run_on_the_host(tmp_path, name, qdiag):
  from oracle_lib import OracleLib
  from rednose_amd.codegen.spec import build_spec
  M, mdl, kw, quat_idx = _wide_model(name)
  mdl = dict(mdl)
  mdl["name"] = name
  spec = build_spec(**mdl, **kw)
  lib, kinds, zmax = _run_host_library(tmp_path, spec)
  o = OracleLib(name)
  D, E = spec.dim_x, spec.dim_err
  rng = np.random.default_rng(E + qdiag)
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  assert not qdiag or np.count_nonzero(Q - np.diag(np.diag(Q))) == 0
  T = 3 * len(kinds)
  sched = np.array([kinds[t % len(kinds)].kind for t in range(T)], dtype=np.int32)
  dts = np.array([0.0 if t % 3 == 1 else 0.01 for t in range(T)])       # a second observation at the same time stamp every third step
  Rt = np.zeros((T, zmax * zmax))
  zs = np.zeros((T, zmax))
  x_init = np.asarray(M.initial_x, dtype=np.float64)
  # away from the initial state itself: live's speed observation has the Jacobian v / |v|, undefined at rest
  x0 = x_init + rng.normal(size=D) * 0.01 * np.maximum(1.0, np.abs(x_init)) * (np.abs(x_init) < 10.0)
  if quat_idx >= 0:
    x0[quat_idx:quat_idx + 4] /= np.linalg.norm(x0[quat_idx:quat_idx + 4])
  A = rng.normal(size=(E, E)) * 0.1 * np.sqrt(M.initial_P_diag)[:, None]
  P0 = np.diag(M.initial_P_diag) + A @ A.T
  zdim = {k.kind: k.zdim for k in kinds}
  for t, kd in enumerate(sched):
    Z = zdim[int(kd)]
    Rk = np.atleast_2d(M.obs_noise.get(int(kd), 0.01 * np.eye(Z)))
    Rt[t, :Z * Z] = Rk.reshape(-1)
    hx = np.zeros(Z)
    o.call(f"h_{int(kd)}", x0.copy(), np.zeros(4), hx)
    zs[t, :Z] = hx + rng.normal(size=Z) * np.sqrt(np.diag(Rk))
  xr, Pr, zr = x0[None].copy(), P0[None].copy(), zs[:, None, :].copy()
  o.batch_run(sched, dts, xr, Pr, zr, Rt, Q, quat_idx=quat_idx)
  xh, Ph, zh = x0.copy(), P0.copy(), zs.copy()
  fl = np.zeros(T, dtype=np.uint8)
  dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
  lib.host_run.argtypes = [dp, dp, dp, dp, ip, dp, dp, ctypes.c_int, ctypes.POINTER(ctypes.c_ubyte)] + [ctypes.c_int] * 3
  lib.host_run(xh.ctypes.data_as(dp), Ph.ctypes.data_as(dp), Q.ctypes.data_as(dp), Rt.ctypes.data_as(dp), sched.ctypes.data_as(ip),
               dts.ctypes.data_as(dp), zh.ctypes.data_as(dp), T, fl.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)), int(quat_idx >= 0), qdiag, 1)
  assert not fl.any()
  what = f"{name} fused run of {T} steps (qdiag={qdiag})"
  assert_close(xh[None], xr, rtol=1e-9, floor=1e-11, what=what + " x")
  assert_close(Ph.reshape(1, -1), Pr.reshape(1, -1), rtol=1e-8, floor=1e-10, what=what + " P")
  assert_close(zh.reshape(1, -1), zr.reshape(1, -1), rtol=1e-8, atol=1e-10 * max(1.0, np.abs(zs).max()), what=what + " y")


# ---- lane-per-filter KERNELS on the host: tile loops, LDS staging, block indexing of the fused runs, masks, flags ------------------
# The whole text of emit_small.kernels() -- k_predict, k_step_*<DO_PREDICT>, k_run, k_run_blk with the device functions they
# inline -- compiled for the host: a workgroup is 64 threads, threadIdx / blockIdx / gridDim are thread-local, __shared__ arrays are
# function-static (built with -fno-gnu-unique: statics of templates would otherwise be shared by every such library in the process),
# rn::wave_lds_sync() is a barrier, the asynchronous HBM -> LDS tile copy is the synchronous one, and
# __builtin_amdgcn_readlane exchanges through a 64-entry array between two barriers.  Workgroups run one after the other.

_RUN_GRID = r"""
template <class F> static void run_grid(int grid, F body) {
  struct Arg { F* body; int lane, block, grid; };
  for (int b = 0; b < grid; b++) {
    pthread_barrier_init(&g_bar, nullptr, 64);
    pthread_t th[64];
    Arg args[64];
    for (int l = 0; l < 64; l++) {
      args[l] = Arg{&body, l, b, grid};
      pthread_create(&th[l], nullptr, [](void* p) -> void* {
        Arg& a = *static_cast<Arg*>(p);
        threadIdx.x = a.lane; blockIdx.x = a.block; gridDim.x = a.grid;
        (*a.body)();
        return nullptr;
      }, &args[l]);
    }
    for (int l = 0; l < 64; l++) pthread_join(th[l], nullptr);
    pthread_barrier_destroy(&g_bar);
  }
}
"""

_KERNEL_PRELUDE = r"""
#include <cmath>
#include <cstdint>
#include <cstring>
#include <pthread.h>
#include <sched.h>
#define __device__
#define __forceinline__ inline
#define __global__
#define __launch_bounds__(...)
#define __shared__ static
#define RN_LDS_PAD 0
struct double2 { double x, y; };
struct Dim3 { int x; };
static thread_local Dim3 threadIdx, blockIdx, gridDim;
static pthread_barrier_t g_bar;
static int g_xchg[64];
inline int __builtin_amdgcn_readlane(int v, int l) {
  g_xchg[threadIdx.x] = v;
  pthread_barrier_wait(&g_bar);
  const int r = g_xchg[l];
  pthread_barrier_wait(&g_bar);
  return r;
}
inline int __double2hiint(double d) { int64_t b; std::memcpy(&b, &d, 8); return (int)(b >> 32); }
inline int __double2loint(double d) { int64_t b; std::memcpy(&b, &d, 8); return (int)(b & 0xffffffff); }
inline double __hiloint2double(int hi, int lo) { const int64_t b = ((int64_t)hi << 32) | (uint32_t)lo; double d; std::memcpy(&d, &b, 8); return d; }
namespace rn {
constexpr int WAVE = 64;
inline void wave_lds_sync() { pthread_barrier_wait(&g_bar); }
inline void async_wait() {}
inline void pin(double&) {}
inline double fast_recip(const double d) { return 1.0 / d; }
inline double fast_rsqrt(const double a) { return 1.0 / std::sqrt(a); }
inline double safe_recip(const double d) { return 1.0 / d; }
inline double safe_rsqrt(const double a) { return 1.0 / std::sqrt(a); }
"""


@_once
def _kernel_host_library(tmp_path, spec):
  from rednose_amd.codegen import emit_small
  hdr = open(HDR, encoding="utf-8").read()
  names = ("lds_stride", "tile_g2l", "tile_l2g", "lds_to_regs", "regs_to_lds", "spd_factor", "spd_forward", "spd_solve", "ldu_factor", "ldu_forward", "ldu_forward_t", "ldu_solve", "rsqrt_pow", "sincos_fast", "normalize_quat")
  helpers = "\n".join(_function_text(hdr, f) for f in names)
  at = hdr.index("struct TilePrefetch")
  prefetch = hdr[hdr.rfind("template <", 0, at):hdr.index("};", at) + 2]
  text = emit_small.kernels(spec)
  if "void k_run(" not in text:      # the step-at-a-time fused run, shipped only when the blocked kernels do not fit (fallback no_run_blk)
    text += "\n" + emit_small.run_kernel(spec, emit_small.norm_text(spec))
  text = text.replace('asm volatile("" : "+v"(v));', ";")        # pin_i: a register constraint of the device
  D, E = spec.dim_x, spec.dim_err
  zmax = max(k.zdim for k in spec.kinds)
  k0 = spec.kinds[0]
  blk_tr = """
extern "C" __attribute__((visibility("default"))) void host_run_blk_tr(int grid, double* x, double* P, const double* Q, const int32_t* kinds, const double* dts, int64_t T,
    double* z, const double* R, int64_t n, int norm_quats, uint8_t* flags, double* tx, double* tP) {
  run_grid(grid, [&] { k_run_blk_tr(x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, nullptr, tx, tP); });
}""" if "void k_run_blk_tr(" in text else ""
  launch = f"""
{_RUN_GRID}
extern "C" __attribute__((visibility("default"))) void host_run(int blocked, int grid, double* x, double* P, const double* Q, const int32_t* kinds, const double* dts, int64_t T, double* z,
                         const double* R, int64_t n, int norm_quats, uint8_t* flags, double* tx, double* tP) {{
  if (blocked) run_grid(grid, [&] {{ k_run_blk(x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, nullptr); }});
  else run_grid(grid, [&] {{ k_run(x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, tx, tP, nullptr, nullptr); }});
}}
{blk_tr}
extern "C" __attribute__((visibility("default"))) void host_step(int grid, double* x, double* P, double* z, const double* R, int r_per_filter, const double* Q, const double* dt_vec, double dt,
                          int64_t n, int norm_quats, uint8_t* flags, const uint8_t* active) {{
  run_grid(grid, [&] {{ k_step_{k0.kind}<true>(x, P, z, R, r_per_filter, nullptr, Q, dt_vec, dt, n, norm_quats, flags, active); }});
}}
extern "C" __attribute__((visibility("default"))) void host_step_ckpt(int grid, double* x, double* P, double* z, const double* R, const double* Q, const double* dt_vec,
                          int64_t n, uint8_t* flags, double* cx, double* cP, double* cz) {{
  run_grid(grid, [&] {{ k_stepc_{k0.kind}<true>(x, P, z, R, 0, nullptr, Q, dt_vec, 0.0, n, 0, flags, nullptr, cx, cP, cz); }});
}}
"""
  src = "\n".join([_KERNEL_PRELUDE, helpers, prefetch,
                   "template <int EPF> inline void tile_g2l_async(const double* g, int cnt, double* lds, int lane) { tile_g2l<EPF>(g, cnt, lds, lane); }",
                   "}  // namespace rn", text, launch])
  cpp, lib = tmp_path / f"{spec.name}_kernels_host.cpp", tmp_path / f"lib{spec.name}_kernels_host.so"
  cpp.write_text(_fiberize(src) if "#include <pthread.h>" in src else src, encoding="utf-8")
  res = subprocess.run(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fno-gnu-unique", "-fvisibility=hidden", "-Wno-unknown-pragmas", "-Wno-attributes", str(cpp), "-o", str(lib)],
                       capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-4000:]
  return ctypes.CDLL(str(lib)), zmax


@pytest.mark.parametrize("name", ["kinematic", "kinematic6_maha", "rand3"])
def test_lane_per_filter_kernels_on_the_host(tmp_path, name):
  """k_run_blk and k_run (with trace) against the oracle's batch_run and each other: ragged last tile, a grid smaller than the tile
  count (grid-stride loop), schedule lengths around the block size, several kinds, gate flags, guard rows around z and the flags;
  then the step kernel with an `active` mask."""
  from oracle_lib import OracleLib
  from rednose_amd.codegen import emit_small
  from rednose_amd.codegen.spec import build_spec
  M, mdl, kw = _model(name)
  spec = build_spec(**mdl, **kw)
  lib, zmax = _kernel_host_library(tmp_path, spec)
  o = OracleLib(name)
  D, E = spec.dim_x, spec.dim_err
  K = emit_small.run_block(spec)
  rng = np.random.default_rng(E)
  n, grid = 150, 2                                       # three tiles (the last with 22 filters) on two workgroups
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  x_init = np.asarray(getattr(M, "initial_x", np.zeros(D)), dtype=np.float64)
  P_init = np.diag(getattr(M, "initial_P_diag", np.ones(E)))
  kset = [k.kind for k in spec.kinds]
  zdim = {k.kind: k.zdim for k in spec.kinds}
  dp, ip, bp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_ubyte)
  lib.host_run.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, ip, dp, ctypes.c_int64, dp, dp, ctypes.c_int64, ctypes.c_int, bp, dp, dp]
  ptr = lambda a, t=dp: a.ctypes.data_as(t)      # noqa: E731
  for T in (1, K - 1, K, 2 * K + 1):
    if T < 1:
      continue
    sched = np.array([kset[t % len(kset)] for t in range(T)], dtype=np.int32)
    dts = rng.uniform(0.005, 0.03, size=T)
    Rt = np.zeros((T, zmax * zmax))
    for t, kd in enumerate(sched):
      Rk = np.atleast_2d(M.obs_noise[int(kd)])
      Rt[t, :Rk.size] = Rk.reshape(-1)
    x0 = x_init[None] + rng.normal(size=(n, D)) * 0.3
    A = rng.normal(size=(n, E, E)) * 0.2
    P0 = P_init[None] + A @ A.transpose(0, 2, 1)
    if T % 2:       # an asymmetric covariance: the fused runs are specified on (P + P^T) / 2 (include/rednose_amd_filter.h)
      Wk = rng.normal(size=(n, E, E))
      P0 = P0 + 1e-3 * np.abs(P0).max(axis=(1, 2), keepdims=True) * (Wk - Wk.transpose(0, 2, 1))
    zs = rng.normal(size=(T, n, zmax)) * np.where(rng.uniform(size=(T, n, 1)) < 0.2, 40.0, 0.5)
    xr, Pr, zr = x0.copy(), 0.5 * (P0 + P0.transpose(0, 2, 1)), zs.copy()
    fr = np.zeros((T, n), dtype=np.uint8)
    o.batch_run(sched, dts, xr, Pr, zr, Rt, Q, flags=fr)
    got = {}
    for blocked in (1, 0):
      xh, Ph = x0.copy(), P0.copy()
      zg = np.full((T + 2, n, zmax), 777.0); zg[1:T + 1] = zs
      fg = np.full((T + 2, n), 99, dtype=np.uint8)
      tx, tP = np.zeros((T, n, D)), np.zeros((T, n, E, E))
      lib.host_run(blocked, grid, ptr(xh), ptr(Ph), ptr(Q), ptr(sched, ip), ptr(dts), T, ptr(zg[1]), ptr(Rt), n, 0, ptr(fg[1], bp),
                   None if blocked else ptr(tx), None if blocked else ptr(tP))
      what = f"{name} T={T} {'k_run_blk' if blocked else 'k_run'}"
      assert (zg[0] == 777.0).all() and (zg[T + 1] == 777.0).all() and (fg[0] == 99).all() and (fg[T + 1] == 99).all(), what + " guard rows"
      assert np.array_equal(fg[1:T + 1] & 1, fr & 1), what + " flags"
      assert_close(xh, xr, rtol=1e-9, floor=1e-11, what=what + " x")
      assert_close(Ph.reshape(n, -1), Pr.reshape(n, -1), rtol=1e-9, floor=1e-11, what=what + " P")
      for t in range(T):      # y of a kind with Z < zmax: the padding columns pass through
        Z = zdim[int(sched[t])]
        assert_close(zg[1 + t][:, :Z], zr[t][:, :Z], rtol=1e-9, atol=1e-11 * max(1.0, np.abs(zs).max()), what=what + f" y[{t}]")
        assert np.array_equal(zg[1 + t][:, Z:], zs[t][:, Z:]), what + " padding columns"
      if not blocked:
        assert np.array_equal(tx[-1], xh) and np.array_equal(tP[-1], Ph), what + " last trace row"
      got[blocked] = (xh, Ph, zg, fg)
    for a, b in zip(got[1], got[0]):                       # same device functions, same compiler here: identical
      assert np.array_equal(a, b), f"{name} T={T}: blocked and traced kernels differ"
    if name.endswith("maha") and T >= K:
      assert (fr & 1).any() and not (fr & 1).all()
  # step kernel: masked-out filters pass through bit for bit with flag 16, the others match the oracle
  k0 = spec.kinds[0]
  Z = k0.zdim
  R = np.ascontiguousarray(np.atleast_2d(M.obs_noise[k0.kind]), dtype=np.float64)
  x0 = x_init[None] + rng.normal(size=(n, D)) * 0.3
  A = rng.normal(size=(n, E, E)) * 0.2
  P0 = P_init[None] + A @ A.transpose(0, 2, 1)
  Wk = rng.normal(size=(n, E, E))      # asymmetric: the step kernels use both halves of P like the reference (ekf_c.c:24,101,115)
  P0 = P0 + 1e-3 * np.abs(P0).max(axis=(1, 2), keepdims=True) * (Wk - Wk.transpose(0, 2, 1))
  z0 = rng.normal(size=(n, Z))
  act = (rng.uniform(size=n) < 0.6).astype(np.uint8)
  dtv = rng.uniform(0.0, 0.03, size=n)
  xr, Pr, zr = x0.copy(), P0.copy(), z0.copy()
  o.batch_step(k0.kind, xr, Pr, zr, R, Q, dtv)
  xh, Ph, zh, fl = x0.copy(), P0.copy(), z0.copy(), np.full(n, 99, dtype=np.uint8)
  lib.host_step.argtypes = [ctypes.c_int, dp, dp, dp, dp, ctypes.c_int, dp, dp, ctypes.c_double, ctypes.c_int64, ctypes.c_int, bp, bp]
  lib.host_step(grid, ptr(xh), ptr(Ph), ptr(zh), ptr(R), 0, ptr(Q), ptr(dtv), 0.0, n, 0, ptr(fl, bp), ptr(act, bp))
  on = act != 0
  assert np.array_equal(xh[~on], x0[~on]) and np.array_equal(Ph[~on], P0[~on]) and np.array_equal(zh[~on], z0[~on]) and (fl[~on] == 16).all()
  assert_close(xh[on], xr[on], rtol=1e-11, floor=1e-13, what=f"{name} masked step x")
  assert_close(Ph[on].reshape(int(on.sum()), -1), Pr[on].reshape(int(on.sum()), -1), rtol=1e-11, floor=1e-13, what=f"{name} masked step P")
  assert_close(zh[on], zr[on], rtol=1e-11, atol=1e-13 * max(1.0, np.abs(z0).max()), what=f"{name} masked step y")
  # k_stepc: the same step (unmasked here) writing its checkpoint -- the plain step's bits, the observations as they came, the filtered pair
  xp, Pp, zp, flp = x0.copy(), P0.copy(), z0.copy(), np.full(n, 99, dtype=np.uint8)
  lib.host_step(grid, ptr(xp), ptr(Pp), ptr(zp), ptr(R), 0, ptr(Q), ptr(dtv), 0.0, n, 0, ptr(flp, bp), None)
  xc, Pc, zc, flc = x0.copy(), P0.copy(), z0.copy(), np.full(n, 99, dtype=np.uint8)
  cx, cP, cz = np.full((n + 1, D), 7.0), np.full((n + 1, E, E), 7.0), np.full((n + 1, Z), 7.0)
  lib.host_step_ckpt.argtypes = [ctypes.c_int, dp, dp, dp, dp, dp, dp, ctypes.c_int64, bp, dp, dp, dp]
  lib.host_step_ckpt(grid, ptr(xc), ptr(Pc), ptr(zc), ptr(R), ptr(Q), ptr(dtv), n, ptr(flc, bp), ptr(cx), ptr(cP), ptr(cz))
  assert np.array_equal(xc, xp) and np.array_equal(Pc, Pp) and np.array_equal(zc, zp) and np.array_equal(flc, flp), f"{name}: checkpointing step vs plain step"
  assert np.array_equal(cx[:n], xc) and np.array_equal(cP[:n], Pc) and np.array_equal(cz[:n], z0), f"{name}: checkpoint"
  assert (cx[n] == 7.0).all() and (cP[n] == 7.0).all() and (cz[n] == 7.0).all(), f"{name}: checkpoint guard rows"


def test_blocked_traced_run_on_the_host(tmp_path):
  """k_run_blk_tr (the blocked structure writing the filtered trace) bit for bit against the step-at-a-time k_run on ragged tiles and
  schedule lengths around the block size, gate flags included."""
  from rednose_amd.codegen import emit_small
  from rednose_amd.codegen.spec import build_spec
  for name in ("kinematic", "kinematic6_maha"):
    M, mdl, kw = _model(name)
    spec = build_spec(**mdl, **kw)
    lib, zmax = _kernel_host_library(tmp_path, spec)
    D, E = spec.dim_x, spec.dim_err
    K = emit_small.run_block(spec)
    rng = np.random.default_rng(E + 3)
    n, grid = 150, 2
    Q = np.ascontiguousarray(M.Q, dtype=np.float64)
    dp, ip, bp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_ubyte)
    ptr = lambda a, t=dp: a.ctypes.data_as(t)      # noqa: E731
    lib.host_run.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, ip, dp, ctypes.c_int64, dp, dp, ctypes.c_int64, ctypes.c_int, bp, dp, dp]
    lib.host_run_blk_tr.argtypes = [ctypes.c_int, dp, dp, dp, ip, dp, ctypes.c_int64, dp, dp, ctypes.c_int64, ctypes.c_int, bp, dp, dp]
    kd = spec.kinds[0]
    Rk = np.atleast_2d(M.obs_noise[kd.kind])
    for T in (1, K - 1, K, 2 * K + 1):
      sched = np.full(T, kd.kind, dtype=np.int32)
      dts = rng.uniform(0.005, 0.03, size=T)
      Rt = np.tile(Rk.reshape(1, -1), (T, 1))
      x0 = np.asarray(getattr(M, "initial_x", np.zeros(D)))[None] + rng.normal(size=(n, D)) * 0.3
      A = rng.normal(size=(n, E, E)) * 0.2
      P0 = np.diag(getattr(M, "initial_P_diag", np.ones(E)))[None] + A @ A.transpose(0, 2, 1)
      zs = rng.normal(size=(T, n, zmax)) * np.where(rng.uniform(size=(T, n, 1)) < 0.2, 40.0, 0.5)
      out = []
      for which in ("k_run", "k_run_blk_tr"):
        xh, Ph, zh = x0.copy(), P0.copy(), zs.copy()
        fl = np.full((T, n), 99, dtype=np.uint8)
        tx, tP = np.full((T + 2, n, D), 5.0), np.full((T + 2, n, E, E), 5.0)
        if which == "k_run":
          lib.host_run(0, grid, ptr(xh), ptr(Ph), ptr(Q), ptr(sched, ip), ptr(dts), T, ptr(zh), ptr(Rt), n, 0, ptr(fl, bp), ptr(tx[1]), ptr(tP[1]))
        else:
          lib.host_run_blk_tr(grid, ptr(xh), ptr(Ph), ptr(Q), ptr(sched, ip), ptr(dts), T, ptr(zh), ptr(Rt), n, 0, ptr(fl, bp), ptr(tx[1]), ptr(tP[1]))
        assert (tx[0] == 5.0).all() and (tx[T + 1] == 5.0).all() and (tP[0] == 5.0).all() and (tP[T + 1] == 5.0).all()
        out.append((xh, Ph, zh, fl, tx, tP))
      for a, b in zip(*out):
        assert np.array_equal(a, b), f"{name} T={T}"


# ---- lane-group STEP KERNELS on the host (emit_wide2.kernels: k_predict, k_step_*<DO_PREDICT>) -------------------------------------
# The whole kernels -- tile loop, scalar phase lane per filter, groups of filters through the matrix phase, error injection, masks,
# flags -- with a workgroup as 64 threads.  The block copies between HBM and LDS are restated here by what they move (the device
# versions differ in how: 16-byte vectors, asynchronous HBM -> LDS transfers); everything else is the generated text.

_WIDE_COPIES = r"""
template <int MAXD> inline void copy_g2l(const double* g, int nd, double* lds, int lane) { for (int i = lane; i < nd; i += 64) lds[i] = g[i]; }
template <int MAXD, bool NT = false> inline void copy_l2g(double* g, int nd, const double* lds, int lane) { for (int i = lane; i < nd; i += 64) g[i] = lds[i]; }
template <int MAXD> inline void async_copy_g2l(const double* g, int nd, double* lds, int lane) { copy_g2l<MAXD>(g, nd, lds, lane); }
inline int odd_start(const double* g) { return (int)(((uintptr_t)g >> 3) & 1); }
template <int MAXD> inline void async_copy_g2l_any(const double* g, int nd, double* lds, int lane) {      // image shifted by one double for odd starts
  const int sh = odd_start(g);
  for (int i = lane; i < nd; i += 64) lds[sh + i] = g[i];
}
template <int MAXD> inline void copy_l2g_any(double* g, int nd, const double* lds, int sh, int lane) { for (int i = lane; i < nd; i += 64) g[i] = lds[sh + i]; }
inline void sched_barrier_(int) {}
"""


@_once
def _wide_kernel_host_library(tmp_path, spec):
  from rednose_amd.codegen import emit_wide2, tuning
  hdr = open(HDR, encoding="utf-8").read()
  helpers = "\n".join(_function_text(hdr, f) for f in ("spd_factor", "spd_forward", "spd_solve", "ldu_factor", "ldu_forward", "ldu_forward_t", "ldu_solve", "rsqrt_pow", "sincos_fast", "normalize_quat"))
  with tuning.using_model(spec):
    text = emit_wide2.kernels(spec)
    FT = emit_wide2.tile_filters(spec)
  text = re.sub(r'asm volatile\("" : "\+v"\((\w+)\)( :: "memory")?\);', ";", text)
  text = text.replace("__builtin_amdgcn_sched_barrier", "rn::sched_barrier_")
  # the scalar-phase functions run on the lanes that own a filter only; their wave_lds_sync() calls are scheduling boundaries for
  # hipcc (a fence inside one wavefront), not rendezvous points -- as barriers they would wait for lanes that never come
  text = re.sub(r"(__device__ \w+ (?:void|int) scal_\w+\(.*?\n}\n)", lambda m: m.group(1).replace("rn::wave_lds_sync();", ";"), text, flags=re.S)
  kinds = [k for k in spec.kinds if k.He_sym is None and k.ea_sym is None]
  entries = []
  for k in kinds:
    entries.append(f"""
extern "C" __attribute__((visibility("default"))) void host_wide_kernel_{k.kind}(int grid, int do_predict, double* x, double* P, double* z, const double* R,
    int r_per_filter, const double* Q, const double* dt_vec, double dt, int64_t n, int norm_quats, uint8_t* flags, const uint8_t* active) {{
  if (do_predict) run_grid(grid, [&] {{ k_step_{k.kind}<true>(x, P, z, R, r_per_filter, nullptr, Q, dt_vec, dt, n, norm_quats, flags, active); }});
  else run_grid(grid, [&] {{ k_step_{k.kind}<false>(x, P, z, R, r_per_filter, nullptr, nullptr, nullptr, 0.0, n, norm_quats, flags, active); }});
}}
extern "C" __attribute__((visibility("default"))) void host_wide_kernel_ckpt_{k.kind}(int grid, double* x, double* P, double* z, const double* R,
    const double* Q, double dt, int64_t n, int norm_quats, uint8_t* flags, double* cx, double* cP, double* cz) {{
  run_grid(grid, [&] {{ k_stepc_{k.kind}<true>(x, P, z, R, 0, nullptr, Q, nullptr, dt, n, norm_quats, flags, nullptr, cx, cP, cz); }});
}}""")
  prelude = _KERNEL_PRELUDE.replace("inline void pin(double&) {}", "inline void pin(double&) {}\n" + _WIDE_COPIES)
  src = "\n".join([prelude, helpers, "}  // namespace rn", text, _RUN_GRID] + entries)
  cpp, lib = tmp_path / f"{spec.name}_wide_kernels_host.cpp", tmp_path / f"lib{spec.name}_wide_kernels_host.so"
  cpp.write_text(_fiberize(src) if "#include <pthread.h>" in src else src, encoding="utf-8")
  res = subprocess.run(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fno-gnu-unique", "-fvisibility=hidden", "-Wno-unknown-pragmas", "-Wno-attributes",
                        str(cpp), "-o", str(lib)], capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-4000:]
  return ctypes.CDLL(str(lib)), kinds, FT


@pytest.mark.parametrize("name", ["kinematic9", "rand11", "live_maha"])
def test_lane_group_step_kernels_on_the_host(tmp_path, name):
  """k_step_*<true> / <false> of the lane-group family against the oracle: a batch that is not a multiple of the tile (ragged last
  tile, odd filter counts in the last group), fewer workgroups than tiles, per-filter dt, a mask, gate flags."""
  from oracle_lib import OracleLib
  from rednose_amd.codegen.spec import build_spec
  M, mdl, kw, quat_idx = _wide_model(name)
  mdl = dict(mdl)
  mdl["name"] = name
  spec = build_spec(**mdl, **kw)
  lib, kinds, FT = _wide_kernel_host_library(tmp_path, spec)
  o = OracleLib(name)
  D, E = spec.dim_x, spec.dim_err
  rng = np.random.default_rng(E)
  n, grid = 2 * FT + max(1, FT // 2) + (FT > 2), 2               # two full tiles and a ragged third one, on two workgroups
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  x_init = np.asarray(M.initial_x, dtype=np.float64)
  P_init = np.diag(M.initial_P_diag)
  dp, bp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_ubyte)
  ptr = lambda a, t=dp: a.ctypes.data_as(t)      # noqa: E731
  gated = 0
  for k in kinds[:3] + [k_ for k_ in kinds[3:] if k_.maha_test]:      # (every kind runs in the function-level test above)
    Z = k.zdim
    R = np.ascontiguousarray(np.atleast_2d(M.obs_noise.get(k.kind, 0.01 * np.eye(Z))), dtype=np.float64)
    x0 = np.tile(x_init, (n, 1)) + rng.normal(size=(n, D)) * 0.01 * np.maximum(1.0, np.abs(x_init))[None] * (np.abs(x_init)[None] < 10.0)
    A = rng.normal(size=(n, E, E)) * 0.1 * np.sqrt(np.diag(P_init))[None, :, None]
    P0 = P_init[None] + A @ A.transpose(0, 2, 1)
    hx = np.zeros((n, Z))
    for i in range(n):
      xq = x0[i].copy()
      if quat_idx >= 0:
        xq[quat_idx:quat_idx + 4] /= np.linalg.norm(xq[quat_idx:quat_idx + 4])
      o.call(f"h_{k.kind}", xq, np.zeros(4), hx[i])
    far = rng.uniform(size=(n, 1)) < 0.34
    z0 = hx + rng.normal(size=(n, Z)) * np.sqrt(np.diag(R))[None] + far * rng.normal(size=(n, Z)) * 40.0 * np.sqrt(P_init.max())
    fn = getattr(lib, f"host_wide_kernel_{k.kind}")
    fn.argtypes = [ctypes.c_int, ctypes.c_int, dp, dp, dp, dp, ctypes.c_int, dp, dp, ctypes.c_double, ctypes.c_int64, ctypes.c_int, bp, bp]
    for mode in ("scalar dt", "dt = 0", "per-filter dt + mask", "update only"):
      dtv = rng.uniform(0.0, 0.02, size=n)
      act = (rng.uniform(size=n) < 0.6).astype(np.uint8)
      xr, Pr, zr = x0.copy(), P0.copy(), z0.copy()
      fr = np.zeros(n, dtype=np.uint8)
      dt_o = {"scalar dt": 0.01, "dt = 0": 0.0, "per-filter dt + mask": dtv, "update only": 0.0}[mode]
      o.batch_step(k.kind, xr, Pr, zr, R, Q, dt_o, quat_idx=quat_idx, flags=fr, do_predict=mode != "update only")
      xh, Ph, zh, fl = x0.copy(), P0.copy(), z0.copy(), np.full(n, 99, dtype=np.uint8)
      masked = mode == "per-filter dt + mask"
      fn(grid, int(mode != "update only"), ptr(xh), ptr(Ph), ptr(zh), ptr(R), 0, ptr(Q), ptr(dtv) if masked else None,
         0.01 if mode == "scalar dt" else 0.0, n, int(quat_idx >= 0), ptr(fl, bp), ptr(act, bp) if masked else None)
      on = (act != 0) if masked else np.ones(n, dtype=bool)
      what = f"{name} kind {k.kind} {mode}"
      assert np.array_equal(xh[~on], x0[~on]) and np.array_equal(Ph[~on], P0[~on]) and np.array_equal(zh[~on], z0[~on]) and (fl[~on] == 16).all(), what + " masked-out filters"
      assert np.array_equal(fl[on] & 1, fr[on] & 1), what + " gate flags"
      gated += int((fl[on] & 1).sum())
      m = int(on.sum())
      assert_close(xh[on], xr[on], rtol=1e-10, floor=1e-12, what=what + " x")
      assert_close(Ph[on].reshape(m, -1), Pr[on].reshape(m, -1), rtol=1e-9, floor=1e-11, what=what + " P")
      assert_close(zh[on], zr[on], rtol=1e-10, atol=1e-12 * max(1.0, np.abs(z0).max()), what=what + " y")
      if mode == "scalar dt":      # k_stepc: the same step writing its checkpoint -- same bits, the observations as they came, the filtered pair
        fc = getattr(lib, f"host_wide_kernel_ckpt_{k.kind}")
        fc.argtypes = [ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_double, ctypes.c_int64, ctypes.c_int, bp, dp, dp, dp]
        xc, Pc, zc, flc = x0.copy(), P0.copy(), z0.copy(), np.full(n, 99, dtype=np.uint8)
        cx, cP, cz = np.full((n + 1, D), 7.0), np.full((n + 1, E, E), 7.0), np.full((n + 1, Z), 7.0)
        fc(grid, ptr(xc), ptr(Pc), ptr(zc), ptr(R), ptr(Q), 0.01, n, int(quat_idx >= 0), ptr(flc, bp), ptr(cx), ptr(cP), ptr(cz))
        assert np.array_equal(xc, xh) and np.array_equal(Pc, Ph) and np.array_equal(zc, zh) and np.array_equal(flc, fl), what + ": checkpointing step vs plain step"
        assert np.array_equal(cx[:n], xc) and np.array_equal(cP[:n], Pc) and np.array_equal(cz[:n], z0), what + ": checkpoint"
        assert (cx[n] == 7.0).all() and (cP[n] == 7.0).all() and (cz[n] == 7.0).all(), what + ": checkpoint guard rows"
  assert (gated > 0) == (name == "live_maha")


# ---- lane-group FUSED RUN kernel on the host (emit_wide3.kernels: k_run with trace, flags, gate) -----------------------------------------

_WAVE_VOTES = r"""
static int g_vote[64];
inline int host_any(int p) {            // __any: a wavefront-wide OR (every lane calls it)
  g_vote[threadIdx.x] = p != 0;
  pthread_barrier_wait(&g_bar);
  int r = 0;
  for (int i = 0; i < 64; i++) r |= g_vote[i];
  pthread_barrier_wait(&g_bar);
  return r;
}
inline int host_readfirstlane(int v) {  // every lane is active wherever the kernels use it: lane 0's value
  g_xchg[threadIdx.x] = v;
  pthread_barrier_wait(&g_bar);
  const int r = g_xchg[0];
  pthread_barrier_wait(&g_bar);
  return r;
}
#define __any host_any
#define __builtin_amdgcn_readfirstlane host_readfirstlane
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
"""


def _two_wave(text, nw=2):
  """The single-wavefront host prelude / grid runner for a workgroup of `nw` wavefronts (k_run2): 64 nw threads, rn::wave_lds_sync() and the
  wavefront-wide votes / exchanges rendezvous the caller's own wavefront, rn::wg_barrier() all of them."""
  text = text.replace("static pthread_barrier_t g_bar;", "static pthread_barrier_t g_wbar[4], g_wg;").replace("struct Dim3 { int x; };\nstatic thread_local Dim3 threadIdx, blockIdx, gridDim;",
                                                                                                             "struct Dim3 { int x; };\nstatic thread_local Dim3 threadIdx, blockIdx, gridDim, blockDim;")
  text = text.replace("pthread_barrier_wait(&g_bar)", "pthread_barrier_wait(&g_wbar[threadIdx.x >> 6])")
  text = text.replace("static int g_xchg[64];", "static int g_xchg[256];").replace("static int g_vote[64];", "static int g_vote[256];")
  text = text.replace("const int r = g_xchg[l];", "const int r = g_xchg[(threadIdx.x & ~63) + l];").replace("const int r = g_xchg[0];", "const int r = g_xchg[threadIdx.x & ~63];")
  text = text.replace("for (int i = 0; i < 64; i++) r |= g_vote[i];", "for (int i = 0; i < 64; i++) r |= g_vote[(threadIdx.x & ~63) + i];")
  text = text.replace("inline void async_wait() {}", "inline void async_wait() {}\ninline void wg_barrier() { pthread_barrier_wait(&g_wg); }\n"
                      "inline void flag_set(int* f, int v) { pthread_barrier_wait(&g_wbar[threadIdx.x >> 6]); if ((threadIdx.x & 63) == 0) __atomic_store_n(f, v, __ATOMIC_RELEASE); }\n"
                      "inline void flag_wait(int* f, int v) { while (__atomic_load_n(f, __ATOMIC_ACQUIRE) != v) sched_yield(); }")
  text = text.replace("pthread_barrier_init(&g_bar, nullptr, 64);", f"for (int w = 0; w < {nw}; w++) pthread_barrier_init(&g_wbar[w], nullptr, 64); pthread_barrier_init(&g_wg, nullptr, {64 * nw});")
  text = text.replace("pthread_barrier_destroy(&g_bar);", f"for (int w = 0; w < {nw}; w++) pthread_barrier_destroy(&g_wbar[w]); pthread_barrier_destroy(&g_wg);")
  text = text.replace("pthread_t th[64];", "pthread_t th[256];").replace("Arg args[64];", "Arg args[256];").replace("for (int l = 0; l < 64; l++)", f"for (int l = 0; l < {64 * nw}; l++)")
  text = text.replace("threadIdx.x = a.lane;", f"threadIdx.x = a.lane; blockDim.x = {64 * nw};")
  assert "&g_bar" not in text and " g_bar" not in text
  return text


@_once
def _wide_run_kernel_host_library(tmp_path, spec, variant="k_run"):
  from rednose_amd.codegen import emit_run2, emit_wide3, tuning
  hdr = open(HDR, encoding="utf-8").read()
  helpers = "\n".join(_function_text(hdr, f) for f in ("spd_factor", "spd_forward", "spd_solve", "ldu_factor", "ldu_forward", "ldu_forward_t", "ldu_solve", "rsqrt_pow", "sincos_fast", "normalize_quat"))
  with tuning.using_model(spec):
    GL, R, FPW = emit_wide3.layout(spec)
    nw = 1
    if variant == "k_run2":
      assert emit_run2.applicable(spec)
      nw = emit_run2.layout2(spec)[3] + 1
      # (k_run2_tri, the packed-triangle trace variant, rides along where the model has it: its row stores lean on the wavefront's lockstep --
      # every lane stores ALL E entries of its row at the row's packed offset, later stores repair the overlap -- which threads do not have:
      # the macro takes its guarded form here, entry j of row r is stored only for j <= r)
      text = (f"constexpr int GLR = {GL}; constexpr int RPL = {R}; constexpr int FPWR = {FPW};\n#define RN_TRI_ST(p, j, r, v) do {{ if ((j) <= (r)) (p)[j] = (v); }} while (0)\n" +
              re.sub(r"__builtin_amdgcn_s_setprio\(\d+\);", ";", emit_run2.kernels(spec, tri=emit_run2.tri_trace(spec))))
    else:
      text = emit_wide3.kernels(spec)
  text = re.sub(r'asm volatile\("" : "\+v"\((\w+)\)( :: "memory")?\);', ";", text)
  text = text.replace("__builtin_amdgcn_sched_barrier", "rn::sched_barrier_")
  text = re.sub(r"(__device__ \w+ (?:void|int) scal_\w+\(.*?\n}\n)", lambda m: m.group(1).replace("rn::wave_lds_sync();", ";"), text, flags=re.S)
  entry = """
extern "C" __attribute__((visibility("default"))) void host_wide_run(int grid, double* x, double* P, const double* Q, const int32_t* kinds, const double* dts,
    int64_t T, double* z, const double* R, int64_t n, int norm_quats, uint8_t* flags, double* tx, double* tP) {
  run_grid(grid, [&] { KERNEL(x, P, Q, kinds, dts, T, z, R, n, norm_quats, flags, tx, tP, nullptr, nullptr); });
}""".replace("KERNEL", variant)
  if "void k_run2_tri(" in text:
    entry += entry.replace("host_wide_run", "host_wide_run_tri").replace("k_run2(", "k_run2_tri(")
  prelude = _KERNEL_PRELUDE.replace("inline void pin(double&) {}", "inline void pin(double&) {}\n" + _WIDE_COPIES).replace("namespace rn {", _WAVE_VOTES + "namespace rn {", 1)
  grid_text = _RUN_GRID
  if variant == "k_run2":
    prelude, grid_text = _two_wave(prelude, nw), _two_wave(_RUN_GRID, nw)
  src = "\n".join([prelude, helpers, "}  // namespace rn", text, grid_text, entry])
  cpp, lib = tmp_path / f"{spec.name}_{variant}_host.cpp", tmp_path / f"lib{spec.name}_{variant}_host.so"
  cpp.write_text(_fiberize(src) if "#include <pthread.h>" in src else src, encoding="utf-8")
  res = subprocess.run(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fno-gnu-unique", "-fvisibility=hidden", "-Wno-unknown-pragmas", "-Wno-attributes",
                        str(cpp), "-o", str(lib)], capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-4000:]
  return ctypes.CDLL(str(lib)), FPW


@pytest.mark.parametrize("name,variant", [("kinematic9", "k_run"), ("rand24", "k_run"), ("live_maha", "k_run"),
                                          ("rand13", "k_run2"), ("rand17", "k_run2"), ("live_maha", "k_run2")])
def test_lane_group_fused_run_kernel_on_the_host(tmp_path, name, variant):
  _fused_run_host_case(tmp_path, name, variant)


def _fused_run_host_case(tmp_path, name, variant):
  """k_run of the lane-group family (and k_run2, its two-wavefront form: emit_run2.py -- a thread per lane of BOTH wavefronts, the
  workgroup barriers as barriers over all 128), filtered trace and flags included, against the oracle's batch_run: a ragged last tile, fewer
  workgroups than tiles, a schedule mixing every non-feature kind with dt = 0 steps, gated observations, an unknown kind (flag 8,
  observation passes through).  The input covariances are ASYMMETRIC: the fused run is specified on (P + P^T) / 2
  (include/rednose_amd_filter.h), which is what the oracle is given."""
  from oracle_lib import OracleLib
  from rednose_amd.codegen.spec import build_spec
  M, mdl, kw, quat_idx = _wide_model(name)
  mdl = dict(mdl)
  mdl["name"] = name
  spec = build_spec(**mdl, **kw)
  lib, FPW = _wide_run_kernel_host_library(tmp_path, spec, variant)
  o = OracleLib(name)
  D, E = spec.dim_x, spec.dim_err
  kinds = [k for k in spec.kinds if k.He_sym is None and k.ea_sym is None]
  zmax = max(k.zdim for k in spec.kinds)
  zdim = {k.kind: k.zdim for k in kinds}
  rng = np.random.default_rng(E + 1)
  n, grid = 2 * FPW + max(1, FPW // 2), 2
  T = 2 * len(kinds)
  sched = np.array([kinds[t % len(kinds)].kind for t in range(T)], dtype=np.int32)
  dts = np.array([0.0 if t % 3 == 1 else 0.01 for t in range(T)])
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  x_init = np.asarray(M.initial_x, dtype=np.float64)
  P_init = np.diag(M.initial_P_diag)
  x0 = np.tile(x_init, (n, 1)) + rng.normal(size=(n, D)) * 0.01 * np.maximum(1.0, np.abs(x_init))[None] * (np.abs(x_init)[None] < 10.0)
  if quat_idx >= 0:
    x0[:, quat_idx:quat_idx + 4] /= np.linalg.norm(x0[:, quat_idx:quat_idx + 4], axis=1, keepdims=True)
  A = rng.normal(size=(n, E, E)) * 0.1 * np.sqrt(np.diag(P_init))[None, :, None]
  P0s = P_init[None] + A @ A.transpose(0, 2, 1)
  Wk = rng.normal(size=(n, E, E))
  dg = np.sqrt(np.einsum("nii->ni", P0s))
  P0 = P0s + 1e-3 * dg[:, :, None] * dg[:, None, :] * (Wk - Wk.transpose(0, 2, 1))      # what the kernel is handed
  Rt = np.zeros((T, zmax * zmax))
  zs = np.zeros((T, n, zmax))
  for t, kd in enumerate(sched):
    Z = zdim[int(kd)]
    Rk = np.atleast_2d(M.obs_noise.get(int(kd), 0.01 * np.eye(Z)))
    Rt[t, :Z * Z] = Rk.reshape(-1)
    for i in range(n):
      hx = np.zeros(Z)
      o.call(f"h_{int(kd)}", x0[i].copy(), np.zeros(4), hx)
      zs[t, i, :Z] = hx + rng.normal(size=Z) * np.sqrt(np.diag(Rk))
    if next(k_ for k_ in kinds if k_.kind == int(kd)).maha_test:      # outliers where they are rejected: an accepted one would wreck the run
      far = rng.uniform(size=n) < 0.3
      zs[t, far, :Z] += rng.normal(size=(int(far.sum()), Z)) * 40.0 * np.sqrt(P_init.max())
  xr, Pr, zr = x0.copy(), 0.5 * (P0 + P0.transpose(0, 2, 1)), zs.copy()
  fr = np.zeros((T, n), dtype=np.uint8)
  xf, Pf = np.zeros((T, n, D)), np.zeros((T, n, E, E))
  o.batch_run(sched, dts, xr, Pr, zr, Rt, Q, quat_idx=quat_idx, flags=fr, xf=xf, Pf=Pf)
  dp, ip, bp = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_ubyte)
  ptr = lambda a, t=dp: a.ctypes.data_as(t)      # noqa: E731
  lib.host_wide_run.argtypes = [ctypes.c_int, dp, dp, dp, ip, dp, ctypes.c_int64, dp, dp, ctypes.c_int64, ctypes.c_int, bp, dp, dp]
  xh, Ph, zh = x0.copy(), P0.copy(), zs.copy()
  fl = np.full((T, n), 99, dtype=np.uint8)
  tx, tP = np.zeros((T, n, D)), np.zeros((T, n, E, E))
  lib.host_wide_run(grid, ptr(xh), ptr(Ph), ptr(Q), ptr(sched, ip), ptr(dts), T, ptr(zh), ptr(Rt), n, int(quat_idx >= 0), ptr(fl, bp), ptr(tx), ptr(tP))
  assert np.array_equal(fl & 1, fr & 1), f"{name} gate flags"
  assert ((fl & 1).any() and not (fl & 1).all()) == (name == "live_maha")
  assert_close(xh, xr, rtol=1e-8, floor=1e-10, what=f"{name} x")
  assert_close(Ph.reshape(n, -1), Pr.reshape(n, -1), rtol=1e-8, floor=1e-10, what=f"{name} P")
  assert_close(tx.reshape(T * n, -1), xf.reshape(T * n, -1), rtol=1e-8, floor=1e-10, what=f"{name} trace x")
  assert_close(tP.reshape(T * n, -1), Pf.reshape(T * n, -1), rtol=1e-8, floor=1e-10, what=f"{name} trace P")
  for t in range(T):
    Z = zdim[int(sched[t])]
    assert_close(zh[t][:, :Z], zr[t][:, :Z], rtol=1e-8, atol=1e-10 * max(1.0, np.abs(zs).max()), what=f"{name} y[{t}]")
  # a second run of the same launch gives the same bits: no lane of this kernel depends on WHEN another one runs between two fences
  xh1, Ph1, zh1 = x0.copy(), P0.copy(), zs.copy()
  tx1, tP1 = np.zeros((T, n, D)), np.zeros((T, n, E, E))
  lib.host_wide_run(grid, ptr(xh1), ptr(Ph1), ptr(Q), ptr(sched, ip), ptr(dts), T, ptr(zh1), ptr(Rt), n, int(quat_idx >= 0), ptr(fl, bp), ptr(tx1), ptr(tP1))
  assert np.array_equal(xh1, xh) and np.array_equal(Ph1, Ph) and np.array_equal(zh1, zh) and np.array_equal(tP1, tP)
  if hasattr(lib, "host_wide_run_tri"):
    # the packed-triangle trace of the same launch (k_run2_tri): the lower triangles of the full trace, bit for bit; nothing else changes
    TRI = E * (E + 1) // 2
    il = np.tril_indices(E)
    lib.host_wide_run_tri.argtypes = lib.host_wide_run.argtypes
    xh3, Ph3, zh3 = x0.copy(), P0.copy(), zs.copy()
    tx3, tP3 = np.zeros((T, n, D)), np.full((T + 1, n, TRI), 7.0)
    lib.host_wide_run_tri(grid, ptr(xh3), ptr(Ph3), ptr(Q), ptr(sched, ip), ptr(dts), T, ptr(zh3), ptr(Rt), n, int(quat_idx >= 0), ptr(fl, bp), ptr(tx3), ptr(tP3))
    assert np.array_equal(tP3[:T], tP[:, :, il[0], il[1]]) and (tP3[T] == 7.0).all(), f"{name}: packed trace"
    assert np.array_equal(xh3, xh) and np.array_equal(Ph3, Ph) and np.array_equal(zh3, zh) and np.array_equal(tx3, tx)
  else:
    assert variant != "k_run2" or name != "live_maha"
  # an unknown kind: flag 8, state and observation untouched for that step
  sched2 = sched.copy(); sched2[1] = 77
  xh2, Ph2, zh2 = x0.copy(), P0.copy(), zs.copy()
  fl2 = np.zeros((T, n), dtype=np.uint8)
  tx2, tP2 = np.zeros((T, n, D)), np.zeros((T, n, E, E))
  lib.host_wide_run(grid, ptr(xh2), ptr(Ph2), ptr(Q), ptr(sched2, ip), ptr(dts), T, ptr(zh2), ptr(Rt), n, int(quat_idx >= 0), ptr(fl2, bp), ptr(tx2), ptr(tP2))
  assert (fl2[1] == 8).all() and np.array_equal(zh2[1], zs[1]) and np.isfinite(xh2).all()
  # the trace of the step before the unknown kind is complete, and the unknown kind's own row is the state it passed through
  assert np.array_equal(tP2[0], tP[0]) and np.array_equal(tx2[0], tx[0]) and np.isfinite(tP2).all()
  assert np.array_equal(tP2[1], tP2[0]) == (dts[1] == 0.0)


# ---- register-broadcast smoother on the host (emit_rts4.kernel: k_rts4) ----------------------------------------------------------------
# The kernel's cross-lane traffic is (a) LDS, ordered by rn::wave_lds_sync() -- written so that every read of another lane's data has a
# fence between it and the write before AND the overwrite after it, which is what threads and barriers need --, and (b) the three
# row_newbcast macros (v_fmac_f64_dpp / v_mov_b64_dpp on the device), restated here as an exchange through a 64-entry array between two
# barriers: lane L of every 16-lane row feeds the row.  Everything else is the generated text.

_RTS4_HOST = r"""
static double g_bc[64];
inline double host_row_bcast(double v, int L) {
  g_bc[threadIdx.x] = v;
  pthread_barrier_wait(&g_bar);
  const double r = g_bc[(threadIdx.x & ~15) + L];
  pthread_barrier_wait(&g_bar);
  return r;
}
#define RN4_FMAC(acc, src, coef, L)  (acc) = std::fma(host_row_bcast((src), (L)), (coef), (acc))
#define RN4_FNMAC(acc, src, coef, L) (acc) = std::fma(-host_row_bcast((src), (L)), (coef), (acc))
#define RN4_BC(dst, src, L)          (dst) = host_row_bcast((src), (L))
#define RN4_SETTLE()                 do { } while (0)
#define RN_RTS_STAMP(i) do { } while (0)
#define __builtin_amdgcn_s_setprio(x)
using std::max; using std::min; using std::fma;
typedef void* lds_void_ptr_host;
inline void __builtin_amdgcn_global_load_lds(const void* g, void* lds_base, int bytes, int, int) {      // one 16-byte piece per lane: base + lane * 16
  std::memcpy(static_cast<char*>(lds_base) + 16 * threadIdx.x, g, bytes);
}
"""


@_once
def _rts4_host_library(tmp_path, spec):
  from rednose_amd.codegen import emit_rts4, tuning
  from rednose_amd.codegen.emit_common import routine_device_function
  hdr = open(HDR, encoding="utf-8").read()
  helpers = "\n".join(_function_text(hdr, f) for f in ("rsqrt_pow", "sincos_fast", "normalize_quat"))
  with tuning.using_model(spec):
    assert emit_rts4.applicable(spec)
    text = emit_rts4.kernel(spec)
    has_tri = emit_rts4.tri_applicable(spec)
    if has_tri:
      text += "\n" + emit_rts4.kernel(spec, tri=True)
  routines = "\n".join(routine_device_function(r)[0] for r in spec.routines() if r.name in ("err_fun", "inv_err_fun"))
  text = re.sub(r'asm volatile\("" : ((?:"\+v"\(\w+\)(?:, )?)+)\);', ";", text)
  text = text.replace("__builtin_amdgcn_sched_barrier", "rn::sched_barrier_")
  text = text.replace("__attribute__((ext_vector_type(2)))", "__attribute__((vector_size(16), aligned(8)))")      # (records of an odd number of doubles start on odd doubles: the device's 16-byte loads only need dword alignment)
  text = text.replace("(const __attribute__((address_space(1))) void*)", "(const void*)")
  text = re.sub(r"(__device__ \w+ (?:void|int) scal_\w+\(.*?\n}\n)", lambda m: m.group(1).replace("rn::wave_lds_sync();", ";"), text, flags=re.S)
  entry = """
extern "C" __attribute__((visibility("default"))) void host_rts4(int grid, const double* xf, const double* Pf, const double* ts, int64_t T, const double* Q, int64_t n,
    int norm_quats, double* xs, double* Ps, const double* xl, const double* Pl) {
  run_grid(grid, [&] { k_rts4(xf, Pf, ts, T, Q, n, norm_quats, xs, Ps, xl, Pl); });
}"""
  if has_tri:
    entry += entry.replace("host_rts4(", "host_rts4_tri(").replace("k_rts4(", "k_rts4_tri(")
  prelude = _KERNEL_PRELUDE.replace("inline void pin(double&) {}", "inline void pin(double&) {}\n" + _WIDE_COPIES + "inline void* lds_offset_ptr(double* p) { return p; }\n")
  prelude = prelude.replace("namespace rn {", _WAVE_VOTES + _RTS4_HOST + "namespace rn {", 1)
  src = "\n".join([prelude, helpers, "}  // namespace rn", routines, text, _RUN_GRID, entry])
  cpp, lib = tmp_path / f"{spec.name}_rts4_host.cpp", tmp_path / f"lib{spec.name}_rts4_host.so"
  cpp.write_text(_fiberize(src) if "#include <pthread.h>" in src else src, encoding="utf-8")
  res = subprocess.run(["g++", "-O0", "-std=c++17", "-fPIC", "-shared", "-pthread", "-fno-gnu-unique", "-fvisibility=hidden", "-Wno-unknown-pragmas", "-Wno-attributes",
                        "-ffp-contract=off", str(cpp), "-o", str(lib)], capture_output=True, text=True)
  assert res.returncode == 0, res.stderr[-4000:]
  dll = ctypes.CDLL(str(lib))
  fn = dll.host_rts4
  dp = ctypes.POINTER(ctypes.c_double)
  fn.argtypes = [ctypes.c_int, dp, dp, dp, ctypes.c_int64, dp, ctypes.c_int64, ctypes.c_int, dp, dp, dp, dp]
  fn.tri = None
  if has_tri:
    fn.tri = dll.host_rts4_tri
    fn.tri.argtypes = fn.argtypes
  return fn


@pytest.mark.timeout(900, method="thread")
def test_register_broadcast_smoother_on_the_host_live(tmp_path):
  """k_rts4 of the live model, whole kernel, against the reference's own rts_smooth (tests/golden/live_rts.npz, produced by running the
  reference class): one full tile and a ragged one, quaternion renormalisation, the recursion started from the recomputed predicted pair."""
  from conftest import golden
  from rednose_amd.codegen.spec import build_spec
  M, mdl, kw, _ = _wide_model("live")
  spec = build_spec(**dict(mdl), **kw)
  fn = _rts4_host_library(tmp_path, spec)
  g = golden("live_rts.npz")
  # the LAST eight estimates of the golden trajectory: a backward recursion's values there depend on nothing older (62 emulated steps take minutes)
  T0, n = len(g["t"]) - 8, 6
  T = 8
  xf = np.ascontiguousarray(np.tile(g["xk_k"][T0:, None, :], (1, n, 1)))
  Pf = np.ascontiguousarray(np.tile(g["Pk_k"][T0:, None], (1, n, 1, 1)))
  # the contract of batch_rts: the LOWER triangle of every covariance is read.  Filter 1 gets garbage above the diagonal.
  iu = np.triu_indices(22, 1)
  Pf[:, 1, iu[0], iu[1]] = 1e30
  xs, Ps = np.full((T + 2, n, 23), 7.0), np.full((T + 2, n, 22, 22), 7.0)
  ts = np.ascontiguousarray(g["t"][T0:], dtype=np.float64)
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  dp = ctypes.POINTER(ctypes.c_double)
  ptr = lambda a: a.ctypes.data_as(dp)      # noqa: E731
  fn(1, ptr(xf), ptr(Pf), ptr(ts), T, ptr(Q), n, 3, ptr(xs[1:]), ptr(Ps[1:]), None, None)
  assert (xs[0] == 7.0).all() and (xs[T + 1] == 7.0).all() and (Ps[0] == 7.0).all() and (Ps[T + 1] == 7.0).all()      # guard rows
  X, P = xs[1:T + 1], Ps[1:T + 1]
  sel = [(i, int(k) - T0) for i, k in enumerate(g["Ps_smooth_idx"]) if int(k) >= T0]
  assert len(sel) >= 2
  for j in range(n):
    assert_close(X[:, j], g["xs_smooth"][T0:], rtol=1e-8, floor=1e-8, what=f"smoothed live states, filter {j}")
    if j != 1:
      assert_close(np.stack([P[k, j] for _, k in sel]).reshape(len(sel), -1), np.stack([g["Ps_smooth"][i] for i, _ in sel]).reshape(len(sel), -1),
                   rtol=1e-8, floor=1e-8, what=f"smoothed live covs, filter {j}")
  # filter 1: what it wrote below the diagonal is what the others wrote (its garbage was never read), above it the garbage plus the correction
  il = np.tril_indices(22)
  assert_close(P[:T - 1, 1][:, il[0], il[1]], P[:T - 1, 0][:, il[0], il[1]], rtol=1e-12, floor=1e-14, what="lower triangle of the filter with garbage above the diagonal")
  qn = np.linalg.norm(X[1:, 0, 3:7], axis=1)
  assert np.abs(qn - 1).max() < 1e-14
  # k_rts4_tri: the same recursion on PACKED lower triangles (what batch_run_tri writes) -- the lower triangles of the result above, and the
  # same states; also in place, and with the newest pair passed in
  assert fn.tri is not None
  il = np.tril_indices(22)
  Pt = np.ascontiguousarray(Pf[:, :, il[0], il[1]])
  xs3, Ps3 = np.full((T + 2, n, 23), 7.0), np.full((T + 2, n, 253), 7.0)
  fn.tri(1, ptr(xf), ptr(Pt), ptr(ts), T, ptr(Q), n, 3, ptr(xs3[1:]), ptr(Ps3[1:]), None, None)
  assert (Ps3[0] == 7.0).all() and (Ps3[T + 1] == 7.0).all() and (xs3[0] == 7.0).all() and (xs3[T + 1] == 7.0).all()
  assert_close(Ps3[1:T + 1].reshape(T * n, -1), P[:, :, il[0], il[1]].reshape(T * n, -1), rtol=1e-13, floor=1e-15, what="packed smoothed covariances vs the full kernel's lower triangles")
  assert_close(xs3[1:T + 1].reshape(T * n, -1), X.reshape(T * n, -1), rtol=1e-13, floor=1e-15, what="packed kernel: smoothed states")
  m = 3                                     # (one ragged tile is enough here: every emulated step of a tile costs seconds)
  xf3, Pf3, Pt3 = np.ascontiguousarray(xf[:, :m]), np.ascontiguousarray(Pf[:, :m]), np.ascontiguousarray(Pt[:, :m])
  xl = np.ascontiguousarray(xf3[T - 1] + 1e-3)
  Pl = np.ascontiguousarray(Pf3[T - 1] * 1.01)
  Pl[1][np.triu_indices(22, 1)] = 0.0      # (the full kernel reads lower triangles: whatever is above is irrelevant)
  xs4, Ps4 = np.zeros((T, m, 23)), np.zeros((T, m, 22, 22))
  fn(1, ptr(xf3), ptr(Pf3), ptr(ts), T, ptr(Q), m, 3, ptr(xs4), ptr(Ps4), ptr(xl), ptr(Pl))
  Xi, Pi = xf3.copy(), Pt3.copy()
  fn.tri(1, ptr(Xi), ptr(Pi), ptr(ts), T, ptr(Q), m, 3, ptr(Xi), ptr(Pi), ptr(xl), ptr(np.ascontiguousarray(Pl[:, il[0], il[1]])))
  keep = [j for j in range(m) if j != 1]
  assert_close(Pi[:, keep].reshape(T * len(keep), -1), Ps4[:, keep][:, :, il[0], il[1]].reshape(T * len(keep), -1), rtol=1e-13, floor=1e-15, what="packed, in place, newest pair passed in")
  assert_close(Xi.reshape(T * m, -1), xs4.reshape(T * m, -1), rtol=1e-13, floor=1e-15, what="packed, in place: states")


@pytest.mark.timeout(900, method="thread")
@pytest.mark.parametrize("name", ["rand8", "rand17"])      # (randz10 and rand11 run on the GPU: tests/test_gpu_random.py::test_smoother_many_shapes; here they cost 20 s of the CPU suite)
def test_register_broadcast_smoother_on_the_host_one_row_per_lane(tmp_path, name):
  """k_rts4 on the small test models -- ONE row slot (8 error states) and two (17, an odd count) --: a numpy restatement of ekf_sym.py:651-690 on the
  oracle's f / F, every filter and step; the newest pair passed in (x_last, P_last) and recomputed; in place (Ps == Pf).  Two of the
  time differences are exactly 0: those steps take the identity-gain path (Ck = I), which the restatement's np.linalg.solve reproduces to
  rounding.  The odd state counts have records of an odd number of doubles: the ragged second tile (3 of 7 filters) ends on a lone double."""
  from oracle_lib import OracleLib
  from rednose_amd.codegen.spec import build_spec
  import examples.random_kf as R
  M = getattr(R, {"rand8": "Random8Kalman", "randz10": "RandomWideObs10Kalman", "rand11": "Random11Kalman", "rand17": "Random17Kalman"}[name])
  spec = build_spec(**M.model())
  fn = _rts4_host_library(tmp_path, spec)
  o = OracleLib(M.name)
  D = spec.dim_x
  rng = np.random.default_rng(D)
  n, T = 7, 6
  X = M.initial_x[None, None] + rng.normal(size=(T, n, D)) * 0.3
  A = rng.normal(size=(T, n, D, D)) * 0.2
  P = np.diag(M.initial_P_diag)[None, None] + A @ A.transpose(0, 1, 3, 2)
  dts = rng.uniform(0.005, 0.03, size=T)
  dts[2] = dts[4] = 0.0                      # ts[1] == ts[2], ts[3] == ts[4]: backward steps k = 1 and k = 3 have dt = 0 (k = T - 2 = 4 is the recursion's first: full path)
  ts = np.cumsum(dts)
  assert spec.identity_at_dt0()
  Q = np.ascontiguousarray(M.Q, dtype=np.float64)
  dp = ctypes.POINTER(ctypes.c_double)
  ptr = lambda a: a.ctypes.data_as(dp)      # noqa: E731

  def reference(xl, Pl):
    xs, Ps = X.copy(), P.copy()
    for j in range(n):
      x1n = P1n = None
      for k in range(T - 2, -1, -1):
        dt = ts[k + 1] - ts[k]
        x1k = np.zeros(D); Fk = np.zeros(D * D)
        o.call("f_fun", X[k, j].copy(), float(dt), x1k); o.call("F_fun", X[k, j].copy(), float(dt), Fk)
        Fk = Fk.reshape(D, D)
        Pkk = np.tril(P[k, j]) + np.tril(P[k, j], -1).T
        P1k = Fk @ Pkk @ Fk.T + dt * Q
        if k == T - 2:
          x1n, P1n = (x1k.copy(), P1k.copy()) if xl is None else (xl[j].copy(), np.tril(Pl[j]) + np.tril(Pl[j], -1).T)
          xs[T - 1, j], Ps[T - 1, j] = x1n, (P1k if Pl is None else Pl[j])
        Ck = np.linalg.solve(P1k, Fk @ Pkk.T).T
        xs[k, j] = X[k, j] + Ck @ (x1n - x1k)
        Ps[k, j] = P[k, j] + Ck @ (P1n - P1k) @ Ck.T
        x1n, P1n = xs[k, j].copy(), np.tril(Ps[k, j]) + np.tril(Ps[k, j], -1).T
    return xs, Ps
  xl = X[T - 1] + rng.normal(size=(n, D)) * 0.01
  Bl = rng.normal(size=(n, D, D)) * 0.1
  Pl = P[T - 1] + Bl @ Bl.transpose(0, 2, 1)
  for tag, a_xl, a_Pl in (("recomputed newest pair", None, None), ("newest pair passed in", xl, Pl)):
    xr, Pr = reference(a_xl, a_Pl)
    xs, Ps = np.full((T + 2, n, D), 7.0), np.full((T + 2, n, D, D), 7.0)
    fn(2, ptr(np.ascontiguousarray(X)), ptr(np.ascontiguousarray(P)), ptr(ts), T, ptr(Q), n, 0, ptr(xs[1:]), ptr(Ps[1:]),
       None if a_xl is None else ptr(np.ascontiguousarray(a_xl)), None if a_Pl is None else ptr(np.ascontiguousarray(a_Pl)))
    assert (xs[0] == 7.0).all() and (xs[T + 1] == 7.0).all() and (Ps[0] == 7.0).all() and (Ps[T + 1] == 7.0).all()
    assert_close(xs[1:T + 1].reshape(T * n, -1), xr.reshape(T * n, -1), rtol=1e-9, floor=1e-11, what=f"{name} smoothed x ({tag})")
    assert_close(Ps[1:T + 1].reshape(T * n, -1), Pr.reshape(T * n, -1), rtol=1e-8, floor=1e-10, what=f"{name} smoothed P ({tag})")
  # in place: outputs aliased onto the inputs give the same bits
  Xi, Pi = np.ascontiguousarray(X.copy()), np.ascontiguousarray(P.copy())
  fn(1, ptr(Xi), ptr(Pi), ptr(ts), T, ptr(Q), n, 0, ptr(Xi), ptr(Pi), ptr(np.ascontiguousarray(xl)), ptr(np.ascontiguousarray(Pl)))
  assert np.array_equal(Xi, xs[1:T + 1]) and np.array_equal(Pi, Ps[1:T + 1])


