"""Stand-in of gs-simp/utils/loss_utils.py: the two names dropin.patch_gs_simp replaces (the launcher never calls these bodies)."""


def l1_loss(network_output, gt):
    return (network_output - gt).abs().mean()


def ssim(img1, img2, window_size=11, size_average=True):
    raise NotImplementedError("stand-in: the launcher test runs the patched ssim")
