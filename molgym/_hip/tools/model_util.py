from molgym_amd.tools.model_util import ModelIO, build_model  # noqa: F401
