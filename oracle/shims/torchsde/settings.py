"""Stand-in for torchsde.settings (test infrastructure; restates torchsde 0.2.5's containers)."""


class _Container:
    """Attribute access + `in` + .all(), which is all sdeint.py:11-15, 840-876, 450-454 need."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    def __contains__(self, item):
        return item in self.__dict__.values()

    def all(self):
        return tuple(self.__dict__.values())

    def __repr__(self):
        return repr(self.all())


METHODS = _Container(euler="euler", milstein="milstein", srk="srk", midpoint="midpoint",
                     reversible_heun="reversible_heun", adjoint_reversible_heun="adjoint_reversible_heun",
                     heun="heun", log_ode_midpoint="log_ode", euler_heun="euler_heun")
NOISE_TYPES = _Container(general="general", diagonal="diagonal", scalar="scalar", additive="additive")
SDE_TYPES = _Container(ito="ito", stratonovich="stratonovich")
LEVY_AREA_APPROXIMATIONS = _Container(none="none", space_time="space-time", davie="davie", foster="foster")
