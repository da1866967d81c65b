from cabinet_amd.models import *  # noqa: F401,F403
