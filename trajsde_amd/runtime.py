"""placeholder, replaced below"""
class NoiseSpec:  # noqa
    pass
class StageRuntime:  # noqa
    def __init__(self, module, stage):
        self.module, self.stage = module, stage
