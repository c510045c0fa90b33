from . import conv, dense  # noqa: F401
from .conv import MessagePassing  # noqa: F401
