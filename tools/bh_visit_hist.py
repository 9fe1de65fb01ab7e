#!/usr/bin/env python3
"""Where do the Barnes-Hut walk's opening tests land in the tree?  CPU only (host octree build of the
library + tools/bh_visit_hist.c).  Prints, for N = 65 536 Plummer, theta = 0.5: visits by depth and by
the body count of the visited node's PARENT (a node is visited iff its parent was opened), and the
share of all visits that a table holding {every node whose parent holds >= S bodies} would serve."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

so = os.path.join(ROOT, "build", "libbh_visit_hist.so")
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-ffp-contract=off",
                           os.path.join(ROOT, "tools", "bh_visit_hist.c"), "-o", so])
hl = C.CDLL(so)
hl.walk_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p]
hl.walk_counts.restype = None

nb = graft.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
theta = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ics = nb.plummer(n)
pos4 = np.concatenate([ics["position"], ics["mass"][:, None]], axis=1).astype(np.float32)
t = nb.host_build_tree(pos4, (0, 0, 0), 64.0, threads=8)
com, w, skip, body = t["com_mass"], t["width"], t["skip"], t["leaf_body"]
m = len(skip)
vis = np.zeros(m, np.uint32)
acc = np.zeros(m, np.uint32)
hl.walk_counts(com.ctypes.data, w.ctypes.data, skip.ctypes.data, m, pos4.ctypes.data, n, theta * theta,
               vis.ctypes.data, acc.ctypes.data)
# parents, depth, bodies per node from the pre-order + skip links
parent = np.full(m, -1, np.int64)
depth = np.zeros(m, np.int64)
stack = []
for i in range(m):
    while stack and skip[stack[-1]] <= i:
        stack.pop()
    if stack:
        parent[i] = stack[-1]
        depth[i] = depth[stack[-1]] + 1
    if skip[i] > i + 1:
        stack.append(i)
leaf = (body >= 0).astype(np.int64)
cs = np.concatenate([[0], np.cumsum(leaf)])
count = cs[skip] - cs[np.arange(m)]
total = int(vis.sum())
print(f"N={n} theta={theta}: nodes {m}, visits {total} ({total / n:.0f}/body), accepted {int(acc.sum())} ({acc.sum() / total:.2%})")
print("depth: nodes, visits share")
for d in range(depth.max() + 1):
    sel = depth == d
    print(f"  {d:2d}: {sel.sum():7d} {vis[sel].sum() / total:7.2%}")
pc = np.where(parent >= 0, count[np.maximum(parent, 0)], n + 1)
print("table = nodes whose parent holds >= S bodies:  S, entries, KiB at 32 B, share of visits")
for S in (16, 32, 64, 128, 256, 512, 1024, 4096):
    sel = pc >= S
    print(f"  {S:5d} {sel.sum():7d} {sel.sum() * 32 / 1024:8.1f} {vis[sel].sum() / total:7.2%}")
order = np.argsort(-vis.astype(np.int64))
cum = np.cumsum(vis[order]) / total
for k in (512, 1024, 2048, 3072, 4096, 5120, 8192, 16384):
    print(f"  hottest {k:6d} nodes ({k * 32 / 1024:.0f} KiB): {cum[k - 1]:.2%} of visits")

# ---- which static proxy picks the hot nodes best?  (share of visits of the top-k nodes by proxy)
gp = np.where(parent >= 0, parent[np.maximum(parent, 0)], -1)
gpc = np.where(gp >= 0, count[np.maximum(gp, 0)], n + 1)
ggp = np.where(gp >= 0, parent[np.maximum(gp, 0)], -1)
ggpc = np.where(ggp >= 0, count[np.maximum(ggp, 0)], n + 1)
def share(score, k):
    o = np.argsort(-score, kind="stable")[:k]
    return vis[o].sum() / total
print("top-k by proxy: k, parent count, grandparent count, great-grandparent count, pc*gpc, oracle")
for k in (1024, 2048, 2560, 4096):
    print(f"  {k:5d} {share(pc.astype(np.float64), k):7.2%} {share(gpc.astype(np.float64), k):7.2%} {share(ggpc.astype(np.float64), k):7.2%} "
          f"{share(pc.astype(np.float64) * gpc, k):7.2%} {share(vis.astype(np.float64), k):7.2%}")
print("threshold on grandparent count: S, entries, share of visits")
for S in (256, 384, 512, 768, 1024, 1536, 2048, 3072, 4096):
    sel = gpc >= S
    print(f"  {S:5d} {sel.sum():7d} {vis[sel].sum() / total:7.2%}")

# ---- a wave-cooperative walk steps through the UNION of its bodies' node sequences: how long is it?
hl.union_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int,
                            C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
order_arr = np.ascontiguousarray(t["order"], np.int32)
print("group of g tree-order bodies: union of visited nodes per group / mean visits per body / mean of the group's longest walk")
for gsz in (8, 16, 32, 64, 256):
    u, sm, mx = C.c_int64(), C.c_int64(), C.c_int64()
    hl.union_counts(com.ctypes.data, w.ctypes.data, skip.ctypes.data, m, pos4.ctypes.data, order_arr.ctypes.data, n, theta * theta, gsz,
                    C.byref(u), C.byref(sm), C.byref(mx))
    ng = (n + gsz - 1) // gsz
    print(f"  g={gsz:3d}: union {u.value / ng:8.0f}   per-body {sm.value / n:7.0f}   longest {mx.value / ng:7.0f}   union/per-body {u.value / ng / (sm.value / n):.2f}")

hl.coop_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                           C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
print("wave-cooperative walk, 64 bodies per wave: K segments, window W, jump to min(resume): iterations / dead / window loads per group, longest wave")
for K in (1, 8):
    for W in (32, 64, 128):
        for um in (0, 1):
            a, d, r, mx = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
            hl.coop_counts(com.ctypes.data, w.ctypes.data, skip.ctypes.data, m, pos4.ctypes.data, order_arr.ctypes.data, n, theta * theta, 64, W, K, um,
                           C.byref(a), C.byref(d), C.byref(r), C.byref(mx))
            ng = (n + 63) // 64
            print(f"  K={K} W={W:3d} min-resume={um}: iterations {a.value / ng:7.0f}  dead {d.value / ng:6.0f}  window loads {r.value / ng:6.0f}  longest wave {mx.value}")

hl.coop_items.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int,
                          C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p]
print("cooperative walk, items = fine segments within R of the group's own place + doubling runs beyond: Kf, R: items/group, iterations/group, longest item, items by iterations (bins of 64)")
for Kf, R, cap in ((384, 8, 1 << 30), (384, 8, 32), (384, 8, 16), (384, 8, 8), (768, 16, 16), (768, 16, 8), (192, 4, 4), (192, 4, 8)):
    hl.set_run_cap(cap)
    a, it, mx = C.c_int64(), C.c_int64(), C.c_int64()
    hist = np.zeros(64, np.int32)
    hl.coop_items(com.ctypes.data, w.ctypes.data, skip.ctypes.data, m, pos4.ctypes.data, order_arr.ctypes.data, n, theta * theta, Kf, R,
                  C.byref(a), C.byref(it), C.byref(mx), hist.ctypes.data)
    ng = (n + 63) // 64
    nz = np.nonzero(hist)[0].max() + 1
    print(f"  Kf={Kf:4d} R={R:2d} cap={min(cap, 9999):4d}: items/group {a.value / ng:5.1f}  iterations/group {it.value / ng:6.0f}  longest {mx.value:5d}  hist {hist[:nz].tolist()}")

hl.coop_window_policy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
print("cooperative walk window policy (per group of 64 bodies): wmin wmax grow shrink: iterations, fills, records fetched, KiB, records per visited node")
for wmin, wmax, gd, sd in ((64, 64, 1, 1), (32, 32, 1, 1), (16, 16, 1, 1), (8, 8, 1, 1), (4, 4, 1, 1), (2, 2, 1, 1), (1, 1, 1, 1),
                           (4, 64, 2, 4), (4, 64, 2, 8), (8, 64, 2, 4), (8, 64, 4, 8), (4, 32, 2, 4), (2, 64, 2, 4)):
    a, f, r = C.c_int64(), C.c_int64(), C.c_int64()
    hl.coop_window_policy(com.ctypes.data, w.ctypes.data, skip.ctypes.data, m, pos4.ctypes.data, order_arr.ctypes.data, n, theta * theta,
                          wmin, wmax, gd, sd, C.byref(a), C.byref(f), C.byref(r))
    ng = (n + 63) // 64
    print(f"  {wmin:3d} {wmax:3d} {gd} {sd}: iterations {a.value / ng:6.0f}  fills {f.value / ng:6.0f}  records {r.value / ng:7.0f}  {r.value / ng * 32 / 1024:7.1f} KiB  {r.value / a.value:5.2f} per visit")

hl.block_walk_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                 C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
bl, ch, ms, lt = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
hl.block_walk_counts(com.ctypes.data, w.ctypes.data, skip.ctypes.data, m, pos4.ctypes.data, order_arr.ctypes.data, n, theta * theta,
                     C.byref(bl), C.byref(ch), C.byref(ms), C.byref(lt))
ng = (n + 63) // 64
print(f"block walk (children stored contiguously), per group of 64 bodies: blocks fetched {bl.value / ng:.0f}, children tested {ch.value / ng:.0f} "
      f"({ch.value / bl.value:.2f} per block), lane tests {lt.value / n:.0f} per body, deepest stack {ms.value} entries")
