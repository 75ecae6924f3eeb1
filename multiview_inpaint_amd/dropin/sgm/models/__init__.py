from .autoencoder import AutoencodingEngine  # noqa: F401
