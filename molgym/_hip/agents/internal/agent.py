from molgym_amd.agents.internal import SchNetAC  # noqa: F401
