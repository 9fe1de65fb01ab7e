import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
nb = g.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
a = nb.plummer(n)
pos4 = np.concatenate([a["position"], a["mass"][:, None]], axis=1).astype(np.float32)
for thr in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,4,8,14,16,24,32").split(",")]:
    nn = C.c_size_t(0); c = (C.c_float * 3)(0, 0, 0)
    ts = []
    for _ in range(15):
        t0 = time.perf_counter()
        nb.lib.nbody_host_build_tree(pos4.ctypes.data, n, c, 64.0, thr, None, None, None, None, None, 0, C.byref(nn))
        ts.append((time.perf_counter() - t0) * 1e3)
    print("threads", thr, "min %.3f ms" % min(ts), "median %.3f" % sorted(ts)[7], flush=True)
