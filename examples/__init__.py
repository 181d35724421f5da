"""Model definitions written against the rednose_amd filter API, and a helper that generates + builds them.

GENERATED_DIR plays the role of the reference's examples/generated/ (produced there by SCons,
/root/reference/examples/SConscript:5-19).
"""
import os

GENERATED_DIR = os.path.abspath(os.environ.get("RN_GEN_DIR") or os.path.join(os.path.dirname(__file__), '..', 'generated'))


def model_table():
  from examples.kinematic_kf import KinematicKalman
  from examples.kinematic6_kf import Kinematic6Kalman
  from examples.kinematic9_kf import Kinematic9Kalman
  from examples.attitude_kf import AttitudeKalman
  from examples.feature_kf import FeatureKalman, WideFeatureKalman
  from examples.live_kf import LiveKalman, ObservationKind as LK
  return {
    "kinematic": lambda d: KinematicKalman.generate_code(d),
    "kinematic6": lambda d: Kinematic6Kalman.generate_code(d),
    "kinematic6_maha": lambda d: _renamed(Kinematic6Kalman, "kinematic6_maha", d, maha_test_kinds=[1]),
    "kinematic9": lambda d: Kinematic9Kalman.generate_code(d),
    "attitude": lambda d: AttitudeKalman.generate_code(d),
    "feature": lambda d: FeatureKalman.generate_code(d),
    "feature36": lambda d: WideFeatureKalman.generate_code(d),
    "live": lambda d: LiveKalman.generate_code(d),
    **{f"rand{n}": (lambda d, n=n: _random(n).generate_code(d)) for n in _random_sizes()},
    **{f"randaff{n}": (lambda d, n=n: _random(n, affine=True).generate_code(d)) for n in _affine_sizes()},
    **{f"randz{n}": (lambda d, n=n: _random(n, wide_obs=True).generate_code(d)) for n, _ in _wide_obs()},
    "rand13_maha": lambda d: _renamed(_random(13), "rand13_maha", d, maha_test_kinds=[1, 3]),
    "live_maha": lambda d: LiveKalman.generate_code(d, name="live_maha", maha_test_kinds=[LK.ECEF_POS]),
  }


def _random_sizes():
  from examples.random_kf import SIZES
  return SIZES


def _affine_sizes():
  from examples.random_kf import AFFINE_SIZES
  return AFFINE_SIZES


def _wide_obs():
  from examples.random_kf import WIDE_OBS
  return WIDE_OBS


def _random(n, affine=False, wide_obs=False):
  import examples.random_kf as R
  return getattr(R, f"RandomWideObs{n}Kalman" if wide_obs else (f"RandomAffine{n}Kalman" if affine else f"Random{n}Kalman"))


def _renamed(cls, name, folder, **kw):
  from rednose_amd.helpers.ekf_sym import gen_code
  mdl = cls.model()
  mdl["name"] = name
  gen_code(folder, **mdl, **kw)


_ENSURED = set()      # (name, folder, tuning) checked in THIS process
_INPUTS = {}          # tuning key -> digest of everything a generated library depends on


def inputs_digest():
  """Digest of everything the text and the build of a model's library depend on: the emitters, the runtime headers, the build module, the helpers
  (gen_code, KalmanFilter.generate_code), the model definitions, sympy's version and the tuning environment.  A library stamped with it ({name}.inputs) is up to date without re-emitting the model --
  the comparison gen_code itself makes needs the emitted text, i.e. 23 s of sympy for the live model, once per process and model."""
  key = (os.environ.get("RN_TUNE", ""), os.environ.get("RN_HIPCC_FLAGS", ""), os.environ.get("RN_ALLOW_SPILLS", ""))
  if key not in _INPUTS:
    import glob
    import hashlib
    import sympy
    repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    files = sorted(glob.glob(os.path.join(repo, "rednose_amd", "codegen", "*.py")) + glob.glob(os.path.join(repo, "rednose_amd", "templates", "*.h")) +
                   glob.glob(os.path.join(repo, "examples", "*.py")) +
                   glob.glob(os.path.join(repo, "rednose_amd", "helpers", "*.py")) + [os.path.join(repo, "rednose_amd", "build.py")])
    h = hashlib.sha256(repr((key, sympy.__version__)).encode())
    for fn in files:
      h.update(os.path.relpath(fn, repo).encode())
      with open(fn, "rb") as f:
        h.update(f.read())
    _INPUTS[key] = h.hexdigest()
  return _INPUTS[key]


def ensure_generated(names=None, folder=GENERATED_DIR):
  """Generate + compile the named example filters (skips up-to-date ones).  Returns the folder.
  A model is checked once per process, folder and tuning: the check itself re-emits the model's source to compare digests -- 15 s of
  sympy work for the live model, which the GPU test suite used to repeat two dozen times (3 of its 11 minutes)."""
  table = model_table()
  for n in (names or table.keys()):
    key = (n, os.path.abspath(folder), os.environ.get("RN_TUNE", ""), os.environ.get("RN_HIPCC_FLAGS", ""), os.environ.get("RN_ALLOW_SPILLS", ""))
    if key in _ENSURED:
      continue
    stamp = os.path.join(folder, f"{n}.inputs")
    fresh = False
    if os.path.exists(stamp) and os.path.exists(os.path.join(folder, f"lib{n}.so")) and os.path.exists(os.path.join(folder, f"{n}.digest")):
      with open(stamp, encoding="utf-8") as f:
        fresh = f.read().strip() == inputs_digest()
    if not fresh:
      table[n](folder)
      with open(stamp, "w", encoding="utf-8") as f:
        f.write(inputs_digest())
    _ENSURED.add(key)
  return folder


def ensure_generated_parallel(names=None, exact_names=(), workers=None):
  """ensure_generated(names) + ensure_exact(exact_names) with the models spread over worker PROCESSES (one model is one sympy emission + one hipcc
  run, independent of the others: the libraries of a fresh tree build in ~3 minutes on 8 cores instead of ~10).  Models a worker has checked are
  marked as checked in this process as well."""
  import subprocess
  import sys
  from concurrent.futures import ThreadPoolExecutor
  names = list(names or model_table().keys())
  cost = {"live": 9, "live_maha": 9, "feature36": 6, "rand56": 5, "rand40": 5, "rand32": 4, "rand24": 4, "feature": 3, "rand17": 3, "rand13": 3, "rand13_maha": 3}
  jobs = sorted([("plain", n) for n in names] + [("exact", n) for n in exact_names], key=lambda j: -cost.get(j[1], 1))      # the long ones first
  workers = workers or max(1, min(6, (os.cpu_count() or 2) - 1))
  repo = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

  def run(job):
    kind, n = job
    call = f"ensure_generated([{n!r}])" if kind == "plain" else f"ensure_exact([{n!r}])"
    res = subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {repo!r}); from examples import ensure_generated, ensure_exact; {call}"],
                         cwd=repo, capture_output=True, text=True)
    if res.returncode != 0:
      raise RuntimeError(f"building {n} ({kind}) failed:\n{res.stdout[-2000:]}\n{res.stderr[-4000:]}")
    return job
  with ThreadPoolExecutor(workers) as ex:
    done = list(ex.map(run, jobs))
  tune, flags, spills = os.environ.get("RN_TUNE", ""), os.environ.get("RN_HIPCC_FLAGS", ""), os.environ.get("RN_ALLOW_SPILLS", "")
  for kind, n in done:
    if kind == "plain":
      _ENSURED.add((n, os.path.abspath(GENERATED_DIR), tune, flags, spills))
    else:
      _ENSURED.add((n, os.path.abspath(EXACT_DIR), ",".join(v for v in (tune, "exact_math=1") if v), flags, spills))
  return GENERATED_DIR


EXACT_DIR = os.path.join(GENERATED_DIR, "exact")      # reference builds with IEEE division / sqrt and the library's sin / cos (tuning knob exact_math)


EXACT_NAMES = ("live", "attitude", "rand5", "rand11", "kinematic9")      # models also built with exact_math=1 by __graft_entry__.build()


def model_class_of(name):
  """The model class behind a name of model_table() (the shipped examples and the seeded random filters)."""
  from examples.kinematic_kf import KinematicKalman
  from examples.kinematic6_kf import Kinematic6Kalman
  from examples.kinematic9_kf import Kinematic9Kalman
  from examples.attitude_kf import AttitudeKalman
  from examples.feature_kf import FeatureKalman, WideFeatureKalman
  from examples.live_kf import LiveKalman
  import examples.random_kf as R
  table = {"kinematic": KinematicKalman, "kinematic6": Kinematic6Kalman, "kinematic9": Kinematic9Kalman, "attitude": AttitudeKalman,
           "feature": FeatureKalman, "feature36": WideFeatureKalman, "live": LiveKalman}
  if name in table:
    return table[name]
  for cls in vars(R).values():
    if isinstance(cls, type) and getattr(cls, "name", None) == name:
      return cls
  raise KeyError(name)


def ensure_exact(names=("live",)):
  """The named filters built with RN_TUNE=exact_math=1 under generated/exact/ (what the fast elementary functions are measured
  against: tests/test_gpu_live.py).  Part of __graft_entry__.build(), so the GPU box finds them prebuilt."""
  old = os.environ.get("RN_TUNE")
  os.environ["RN_TUNE"] = ",".join(v for v in (old, "exact_math=1") if v)
  try:
    return ensure_generated(list(names), folder=EXACT_DIR)
  finally:
    if old is None:
      del os.environ["RN_TUNE"]
    else:
      os.environ["RN_TUNE"] = old
