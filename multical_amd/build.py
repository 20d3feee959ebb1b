"""Build the HIP back-end (libmcba.so) in-tree for gfx950.

    python -m multical_amd.build            # incremental
    python -m multical_amd.build --force

hipcc cross-compiles without a GPU.  The shared object is written to multical_amd/_build/libmcba.so (git-ignored,
but it travels to the GPU box with the repo snapshot).  No CPU variant of the kernels is built: the product path is
HIP-only.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# MCBA_BUILD_VARIANT=name + MCBA_EXTRA_FLAGS="-DX=1 ..." build an experimental variant into _build_name/ (profiling aid:
# several variants of the kernels can then be compared in one GPU session, selected with MCBA_LIB_PATH)
VARIANT = os.environ.get("MCBA_BUILD_VARIANT", "")
OUT = os.path.join(HERE, "_build" + ("_" + VARIANT if VARIANT else ""))
LIB = os.path.join(OUT, "libmcba.so")
ARCH = "gfx950"
SOURCES = ["mcba_api.hip", "mcba_cam_pin4.hip", "mcba_cam_pin5.hip", "mcba_cam_pin8.hip", "mcba_cam_pin12.hip",
           "mcba_cam_pin14.hip", "mcba_cam_fish4.hip", "mcba_cam_mix14.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join("..", "..", "include", "mcba.h")]
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wall", "-Wno-unused-function"] + \
        (["-DMCBA_ENV_SWITCHES=1"] if VARIANT else []) + os.environ.get("MCBA_EXTRA_FLAGS", "").split()
# (-DMCBA_ENV_SWITCHES: only VARIANT builds read the MCBA_* experiment switches from the environment; the product library takes
#  them through mcba_debug_set_switch alone -- csrc/mcba_api.hip: dbg_switch)


def hipcc():
  for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
    if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
      return cand
  raise RuntimeError("hipcc not found")


def _newer(target, deps):
  if not os.path.exists(target):
    return True
  t = os.path.getmtime(target)
  return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
  os.makedirs(OUT, exist_ok=True)
  hdrs = [os.path.join(CSRC, h) for h in HEADERS]
  jobs = []
  for src in SOURCES:
    s = os.path.join(CSRC, src)
    o = os.path.join(OUT, src.replace(".hip", ".o"))
    if force or _newer(o, [s] + hdrs):
      jobs.append((s, o))

  def compile_one(job):
    s, o = job
    cmd = [hipcc()] + FLAGS + ["-c", s, "-o", o]
    r = subprocess.run(cmd, capture_output=True, text=True)
    return job, r

  if jobs:
    if verbose:
      print(f"[mcba build] compiling {len(jobs)} translation unit(s) for {ARCH} ...", flush=True)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
      for (s, o), r in ex.map(compile_one, jobs):
        if r.returncode != 0:
          sys.stderr.write(r.stdout + r.stderr)
          raise RuntimeError(f"hipcc failed on {s}")
        if verbose and r.stderr.strip():
          sys.stderr.write(r.stderr)
  objs = [os.path.join(OUT, src.replace(".hip", ".o")) for src in SOURCES]
  if force or jobs or _newer(LIB, objs):
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      sys.stderr.write(r.stdout + r.stderr)
      raise RuntimeError("link failed")
    if verbose:
      print(f"[mcba build] linked {LIB}", flush=True)
  return LIB


if __name__ == "__main__":
  build(force="--force" in sys.argv)
