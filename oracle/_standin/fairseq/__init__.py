from . import utils, modules, dataclass, models  # noqa
