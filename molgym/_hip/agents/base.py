from molgym_amd.agents.base import AbstractActorCritic  # noqa: F401
