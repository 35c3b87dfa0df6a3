from molgym_amd.agents.dists import GaussianMixtureModel  # noqa: F401
