from molgym_amd.spaces import *  # noqa: F401,F403
from molgym_amd.spaces import ActionSpace, ObservationSpace  # noqa: F401
