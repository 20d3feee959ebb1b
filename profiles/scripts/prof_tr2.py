import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from util import load_golden, mirror
from multical_amd.backend import Handle
for name in ("tiny_rolling", "tiny", "tiny_edge", "cfg1"):
    g, rig = load_golden(name)
    with Handle(mirror(rig)) as h:
        res = h.solve(g["x0"])
        print(name, "nfev", res.nfev, "njev", res.njev, "status", res.status)
