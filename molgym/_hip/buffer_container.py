from molgym_amd.buffer import PPOBufferContainer  # noqa: F401
