"""Shim for the tuning scripts: `knob(nb, "sym_wpb").value = 8` sets that knob on every live Simulation of the mirror `nb` and on
every one created afterwards.  The library keeps its knobs per handle (include/nbody_hip.h nbody_set_tuning; csrc/kernels.h
struct Tuning) -- the scripts were written when they were process globals.  Load the TUNING build for the experimental walks
and the in-kernel stamps: graft.load_package(tuning=True)."""
import weakref

_DEFAULTS = dict(cross_sym=1, sym_packed=1, bf_fast_variant=0, sym_wpb=12, sym_rounds=1, sym_reduce_split=1, cross_slots=3072,
                 cross_ipt=0, cross_wpb=4, bh_walk_split=0, bh_walk_order=1, bh_reduce_split=1, tree_max_tie=64, bh_walk_variant=0,
                 bh_walk_lds_block=1024, bh_hot_cap=2048, bh_walk_debug=0, sym_debug=0)
_live = {}


def _track(nb):
    if id(nb) in _live:
        return _live[id(nb)]
    sims = weakref.WeakSet()
    _live[id(nb)] = sims
    orig = nb.Simulation.__init__

    def init(self, *a, **kw):
        orig(self, *a, **kw)
        sims.add(self)

    nb.Simulation.__init__ = init
    return sims


class knob:
    def __init__(self, nb, name):
        self.nb, self.name = nb, name.replace("nbody_", "")
        self.sims = _track(nb)

    @property
    def value(self):
        return self.nb._default_tuning.get(self.name, _DEFAULTS[self.name])

    @value.setter
    def value(self, v):
        self.nb._default_tuning[self.name] = int(v)
        for s in list(self.sims):
            if s._h:
                s.set_tuning(self.name, int(v))
