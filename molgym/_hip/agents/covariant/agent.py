from molgym_amd.agents.covariant import CovariantAC  # noqa: F401
