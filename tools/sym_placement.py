"""Diagnostic: where the dispatcher put the waves of the symmetric kernel (HW_ID/XCC_ID stamps)."""
import ctypes, os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = 65536
ics = nb.plummer(n)
wps = knob(nb, "sym_wpb")   # waves per workgroup (16, 12 or 8)
dbg = knob(nb, "sym_debug")
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BRUTE_FORCE, math_mode=nb.FAST)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.5)
wps.value, dbg.value = 12, 4
for _ in range(50):
    sim.update_forces()
sim.sync()
nw = 3072
buf = (ctypes.c_ulonglong * (3 * nw))()
assert nb.lib.nbody_sym_read_stamps(buf, nw) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 3)
hw = (st[:, 2] >> np.uint64(16)) & np.uint64(0xFFFFFFFF)
xcc = (st[:, 2] >> np.uint64(48)) & np.uint64(0xF)
chunks = st[:, 2] & np.uint64(0xFFFF)
wave_slot = hw & np.uint64(0xF); simd = (hw >> np.uint64(4)) & np.uint64(3); cu = (hw >> np.uint64(8)) & np.uint64(0xF)
sh = (hw >> np.uint64(12)) & np.uint64(1); se = (hw >> np.uint64(13)) & np.uint64(7)
key_cu = [(int(x), int(e), int(s), int(c)) for x, e, s, c in zip(xcc, se, sh, cu)]
key_simd = [k + (int(d),) for k, d in zip(key_cu, simd)]
per_cu = collections.Counter(key_cu); per_simd = collections.Counter(key_simd)
print("distinct CUs", len(per_cu), "waves per CU histogram", sorted(collections.Counter(per_cu.values()).items()))
print("distinct SIMDs", len(per_simd), "waves per SIMD histogram", sorted(collections.Counter(per_simd.values()).items()))
life = st[:, 1].astype(np.float64) / 100.0
cyc = st[:, 0].astype(np.float64) / (chunks.astype(np.float64) * 64)
bysimd = collections.defaultdict(list)
for k, l, c in zip(key_simd, life, cyc):
    bysimd[k].append((l, c))
for cnt in sorted(set(per_simd.values())):
    ls = [max(x[0] for x in v) for k, v in bysimd.items() if len(v) == cnt]
    cs = [np.mean([x[1] for x in v]) for k, v in bysimd.items() if len(v) == cnt]
    print(f"SIMDs with {cnt} waves: {len(ls)}; last wave ends at median {np.median(ls):.0f} us; mean cycles/step per wave {np.mean(cs):.0f}")
# is the spread systematic?  lifetimes by XCD and by shader engine, and the spread inside one CU
for name, grp in (("XCD", [int(x) for x in xcc]), ("XCD,SE", [(int(x), int(e)) for x, e in zip(xcc, se)])):
    by = collections.defaultdict(list)
    for g, l in zip(grp, life):
        by[g].append(l)
    print(f"lifetime by {name}:", "  ".join(f"{g}: med {np.median(v):.0f} max {np.max(v):.0f}" for g, v in sorted(by.items())))
bycu = collections.defaultdict(list)
for k, l in zip(key_cu, life):
    bycu[k].append(l)
spread_in_cu = [max(v) - min(v) for v in bycu.values()]
cu_max = [max(v) for v in bycu.values()]
print(f"inside a CU: lifetime spread median {np.median(spread_in_cu):.0f} us; across CUs: slowest wave of a CU from {min(cu_max):.0f} to {max(cu_max):.0f} us (median {np.median(cu_max):.0f})")
print("first 16 waves (one workgroup): simd ids", [int(x) for x in simd[:16]], "cu", [int(x) for x in cu[:16]])
dbg.value = 0
