"""Where do the 26 us of k_lsmr_fused2 go?  What-if variants of the kernel (MCBA_BUILD_VARIANT builds with -DMCBA_EXP_F2_*: wrong
results on purpose, the same memory traffic / launch shape otherwise), each timed by mcba_time_lsmr_iteration in its own process:
    NO_TMAT    the 2.3 KB of That per view are not streamed (review item 4b)
    NO_STATE   the forward model + derivatives of an observation are not evaluated (constant state)
    NO_MATH    the two products of an observation are not formed (loads / stores / per-view work stay)
    NO_REDUCE  no wave butterfly, no That^T product, one store per view
    W3 / W4    amdgpu_waves_per_eu(3 | 4): the register allocator spills to fit 168 / 128 VGPRs; grids 3072 / 4096 (and 2048)
  python profiles/scripts/prof_f2_whatif.py [cfg3 cfg4]      (needs multical_amd/_build_f2_*/libmcba.so; built by collect_r06.sh)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CODE = """
import sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')
from multical_amd import synthetic, calibration
from multical_amd.backend import Handle
out = {}
for cfg in %r:
  c = calibration.from_rig(synthetic.make_rig(cfg))
  with Handle(c) as h:
    h.set_lsmr_fused(2)
    h.set_lsmr_grid(%d)
    h.time_lsmr_iteration(c.param_vec, repeats=50)
    out[cfg] = [1e3 * v for v in h.time_lsmr_iteration(c.param_vec, repeats=300)]
print(json.dumps(out))
"""
cfgs = sys.argv[1:] or ["cfg3", "cfg4"]
res = {}
for variant, grid in (("", 2048), ("f2_NO_TMAT", 2048), ("f2_NO_STATE", 2048), ("f2_NO_MATH", 2048), ("f2_NO_REDUCE", 2048),
                      ("f2_W3", 3072), ("f2_W4", 4096), ("f2_W3", 2048), ("f2_W4", 2048)):
  lib = os.path.join(ROOT, "multical_amd", "_build" + ("_" + variant if variant else ""), "libmcba.so")
  if not os.path.exists(lib):
    continue
  env = dict(os.environ, MCBA_LIB_PATH=lib)
  r = subprocess.run([sys.executable, "-c", CODE % (ROOT, ROOT, cfgs, grid)], env=env, capture_output=True, text=True)
  if r.returncode != 0:
    res[(variant or "product") + "@%d" % grid] = r.stderr[-300:]
    continue
  res[(variant or "product") + "@%d" % grid] = json.loads(r.stdout.strip().splitlines()[-1])
print(json.dumps(res, indent=1))
for cfg in cfgs:
  base = res["product@2048"][cfg][0]
  print(cfg, "k_lsmr_fused2 %.1f us;" % base, "; ".join("%s %.1f us (%+.1f)" % (k, v[cfg][0], v[cfg][0] - base) for k, v in res.items() if k != "product@2048" and isinstance(v, dict)))
