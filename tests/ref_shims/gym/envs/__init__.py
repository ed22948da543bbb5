from . import registration
