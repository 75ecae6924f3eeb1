from multiview_inpaint_amd.svd.engine import OPENAIUNETWRAPPER, IdentityWrapper, OpenAIWrapper  # noqa: F401
