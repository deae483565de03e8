"""CPU restatement of the reference's frame metrics.  TEST INFRASTRUCTURE (imported by tests/ only).

Restates ``Evaluator.forward`` (/root/reference/ivideogpt/utils/video_metric.py:63-100) without LPIPS:
  * ``nn.MSELoss(reduction='none')(a, b).mean([1, 2, 3])``                                             (:76)
  * ``piqa.PSNR(epsilon=1e-08, value_range=1.0, reduction='none')``: 10 log10(1 / (mse + eps)) per image (:23, :77)
  * ``piqa.SSIM(window_size=11, sigma=1.5, n_channels=3, reduction='none')``                             (:24, :78):
    Gaussian window exp(-(i - 5)^2 / (2 sigma^2)) normalised to 1, applied separably per channel WITHOUT padding, moments
    mu_x, mu_y, E[x^2] - mu_x^2, E[y^2] - mu_y^2, E[xy] - mu_x mu_y, c1 = 0.01^2, c2 = 0.03^2 (value_range 1),
    ss = (2 mu_xy + c1) / (mu_xx + mu_yy + c1) * (2 sigma_xy + c2) / (sigma_xx + sigma_yy + c2), mean over (C, H-10, W-10)
  * best of t (:88-93): metrics reshaped (t, B, T), mean over the frames, min (mse) / max (psnr, ssim) over t.

``piqa`` (a third-party dependency of the reference, unpinned in its requirements.txt) is absent from the build image and there
is no network: the SSIM / PSNR definitions above restate piqa's published implementation and are PARITY UNPINNED -- no output
of piqa itself could be generated here.  MSE and the best-of-t reduction are the reference's own code.
"""
import torch
import torch.nn.functional as F


def gaussian_window(size=11, sigma=1.5):
    k = torch.arange(size, dtype=torch.float32) - (size - 1) / 2
    k = torch.exp(-(k ** 2) / (2 * sigma ** 2))
    return k / k.sum()


def _filter(x, k):
    """x [N, C, H, W]; separable per-channel valid convolution."""
    C = x.shape[1]
    kh = k.view(1, 1, -1, 1).repeat(C, 1, 1, 1)
    kw = k.view(1, 1, 1, -1).repeat(C, 1, 1, 1)
    return F.conv2d(F.conv2d(x, kh, groups=C), kw, groups=C)


def ssim_per_image(x, y, k=None, k1=0.01, k2=0.03):
    k = gaussian_window() if k is None else k
    c1, c2 = k1 ** 2, k2 ** 2
    mu_x, mu_y = _filter(x, k), _filter(y, k)
    mu_xx, mu_yy, mu_xy = mu_x ** 2, mu_y ** 2, mu_x * mu_y
    s_xx, s_yy, s_xy = _filter(x ** 2, k) - mu_xx, _filter(y ** 2, k) - mu_yy, _filter(x * y, k) - mu_xy
    cs = (2 * s_xy + c2) / (s_xx + s_yy + c2)
    ss = (2 * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs
    return ss.flatten(1).mean(-1)


@torch.no_grad()
def frame_metric_rows(video_1, video_2):
    """video_1 [B, T, 3, H, W] ground truth, video_2 [t*B, T, 3, H, W] predictions -> [B, 3] (mse, psnr, ssim) best of t."""
    B, T, C, H, W = video_1.shape
    t = video_2.shape[0] // B
    v1 = video_1.float().repeat([t, 1, 1, 1, 1]).reshape(-1, C, H, W)
    v2 = video_2.float().reshape(-1, C, H, W)
    mse = ((v1 - v2) ** 2).mean([1, 2, 3])
    psnr = 10 * torch.log10(1.0 / (mse + 1e-8))
    ssim = ssim_per_image(v1, v2)
    return torch.stack([mse.reshape(t, B, T).mean(-1).min(0).values, psnr.reshape(t, B, T).mean(-1).max(0).values,
                        ssim.reshape(t, B, T).mean(-1).max(0).values], 1)
