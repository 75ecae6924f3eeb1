"""MI355X-native hot path of JiuTongBro/MultiView_Inpaint: Gaussian-splat rasterizer-with-depth
and the SVD temporal-UNet denoise loop. See DESIGN.md."""
__version__ = "0.1.0"
