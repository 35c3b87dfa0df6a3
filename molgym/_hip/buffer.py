from molgym_amd.buffer import DynamicPPOBuffer, discount_cumsum  # noqa: F401
