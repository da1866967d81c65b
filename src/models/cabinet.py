from cabinet_amd.models.cabinet import *  # noqa: F401,F403
from cabinet_amd.models import cabinet as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
