from . import _Unavailable


class Polygon(_Unavailable):
    pass
