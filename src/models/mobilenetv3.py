from cabinet_amd.models.mobilenetv3 import *  # noqa: F401,F403
from cabinet_amd.models import mobilenetv3 as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
