from cabinet_amd.models.constants import *  # noqa: F401,F403
from cabinet_amd.models import constants as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
