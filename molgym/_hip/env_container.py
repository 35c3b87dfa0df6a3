from molgym_amd.env_container import AsyncEnvContainer, SimpleEnvContainer, VecEnv  # noqa: F401
