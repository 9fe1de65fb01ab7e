"""Diagnostic: when do the waves of the Barnes-Hut walk start and end inside one launch, and how many
iterations does each run?  (DBG instantiation of k_bh_walk: s_memrealtime stamps per wave.)"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _knobs import knob  # noqa: E402
nb = graft.load_package(tuning=True)   # (the build with the experimental walks and the in-kernel stamps)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
splits = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8]
dbg = knob(nb, "bh_walk_debug")
split = knob(nb, "bh_walk_split")
ics = nb.plummer(n)
sim = nb.Simulation(ics, (0, 0, 0), 64.0, method=nb.BARNES_HUT, math_mode=nb.FAST, tree_build=nb.TREE_DEVICE)
sim.settings = nb.Settings(1.0, 1e-2, 1e-3, 0.25)
for K in splits:
    split.value = K
    dbg.value = 0
    for _ in range(40):
        sim.update_forces()
    sim.sync()
    dbg.value = 1
    sim.update_forces(); sim.sync()
    dbg.value = 0
    Kw = K if K > 0 else max(1, min(32, 8192 // ((n + 63) // 64)))   # 0 = automatic
    nw = min(16384, Kw * ((n + 63) // 64))
    buf = (ctypes.c_ulonglong * (3 * nw))()
    assert nb.lib.nbody_bh_read_stamps(buf, nw) == 0
    raw = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 3)
    st = raw.astype(np.float64)
    st[:, 2] = (raw[:, 2] & np.uint64(0xFFFFF)).astype(np.float64)
    hw = (raw[:, 2] >> np.uint64(20)) & np.uint64(0xFFFFFFFF)
    xcc = ((raw[:, 2] >> np.uint64(52)) & np.uint64(0xF)).astype(np.int64)
    # HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    cu = ((hw >> np.uint64(8)) & np.uint64(0xF)).astype(np.int64); sh = ((hw >> np.uint64(12)) & np.uint64(1)).astype(np.int64); se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
    place = ((xcc * 8 + se) * 2 + sh) * 16 + cu   # a CU
    t0 = st[:, 0].min()
    beg, end, it = (st[:, 0] - t0) / 100.0, (st[:, 1] - t0) / 100.0, st[:, 2]   # us
    span = end.max()
    life = end - beg
    print(f"K={K}: {nw} waves, launch span {span:.1f} us; wave starts: median {np.median(beg):.1f} max {beg.max():.1f} us; "
          f"ends: 25% {np.quantile(end,0.25):.1f} 50% {np.median(end):.1f} 75% {np.quantile(end,0.75):.1f} 95% {np.quantile(end,0.95):.1f} us")
    print(f"      iterations per wave (max lane): median {np.median(it):.0f} 90% {np.quantile(it,0.9):.0f} 99% {np.quantile(it,0.99):.0f} max {it.max():.0f}; "
          f"sum {it.sum():.3e}; cycles per iteration of the longest waves ~ {np.median((life*2400/np.maximum(it,1))[it > np.quantile(it,0.99)]):.0f}")
    alive = [(end > f * span).mean() for f in (0.25, 0.5, 0.75, 0.9)]
    print("      waves still running at 25/50/75/90 % of the span:", " ".join(f"{a*100:.0f}%" for a in alive))
    # which (group block, segment) pairs share a CU?  wave w = (seg * n_blocks + block) * 4 + wave-in-block
    n_blocks = (n + 255) // 256
    wid = np.arange(nw)
    seg_of, blk_of = (wid // 4) // n_blocks, (wid // 4) % n_blocks
    cus = np.unique(place)
    same_seg, spread = [], []
    for c in cus[:64]:
        m = place == c
        segs, blks = seg_of[m], blk_of[m]
        same_seg.append(len(np.unique(segs)))
        spread.append(blks.max() - blks.min())
    print(f"      {len(cus)} CUs seen; per CU: {np.mean([ (place==c).sum() for c in cus]):.1f} waves; distinct segments per CU median {np.median(same_seg):.0f}; "
          f"group-block spread per CU median {np.median(spread):.0f} of {n_blocks}")
    c0 = cus[0]; m = place == c0
    print("      CU", c0, "holds (segment, block):", sorted(set(zip(seg_of[m].tolist(), blk_of[m].tolist())))[:16])
split.value = 0
