"""Frame metrics of predicted clips on the MI355X (libivg ``ivg_frame_metrics``): the reference's ``Evaluator.forward``
(/root/reference/ivideogpt/utils/video_metric.py:63-100) without LPIPS / FVD (external network weights): per-frame MSE,
PSNR and SSIM (piqa semantics), mean over a trajectory's frames, best of the ``t`` samples drawn per trajectory."""
import ctypes as C
import threading

import torch

from . import _lib
from .packing import dtype_code


_WS = {}   # (device, stream) -> scratch tensor of ivg_frame_metrics (partial sums of the metric tiles), least recently used first
_WS_MAX = 16   # a raw stream handle can be reused by a NEW stream after the old one died: a bounded cache also bounds how long such a
               # stale association lives (the buffer is private to one call's kernels either way -- ordered by the stream it runs on)
_WS_LOCK = threading.Lock()   # the lanes' host threads (bench.py --lanes) all come through here: pop / insert / evict as one step


@torch.no_grad()
def frame_metric_rows(video_gt, video_pred, gt_t0=0, pred_t0=0, frames=None):
    """video_gt (B, T, 3, H, W) float32 / bfloat16 on the GPU; video_pred float32 (t * B, T', 3, H, W), sample k of trajectory b
    at row k * B + b.  Frames [gt_t0, gt_t0 + frames) of the ground truth are compared with [pred_t0, pred_t0 + frames) of the
    prediction (default: everything after the offsets).  -> float32 (B, 3) rows (mse, psnr, ssim), best of t per trajectory."""
    lib = _lib.load()
    if not video_gt.is_cuda:
        raise RuntimeError("frame metrics run on the MI355X only (no CPU path)")
    gt = video_gt if video_gt.dtype in (torch.float32, torch.bfloat16) else video_gt.float()
    gt, pred = gt.contiguous(), video_pred.float().contiguous()
    B, Tg, _, H, W = gt.shape
    n, Tp = pred.shape[:2]
    T = frames if frames is not None else min(Tg - gt_t0, Tp - pred_t0)
    rows = torch.empty(B, 3, dtype=torch.float32, device=gt.device)
    nbytes = lib.ivg_frame_metrics_ws_bytes(n, T, H, W)
    # The scratch is kept PER (device, stream): no allocation on the hot path (a serving loop calls this every step), and two batches
    # in flight on two streams / host threads (bench.py --lanes) never share partial sums -- the tile kernel writes them and the reduce
    # kernel of the same call reads them back, ordered by the stream.  A buffer that has to grow is replaced by one allocated under
    # the same current stream, so the caching allocator hands the old block only to later work of that stream.
    stream = torch.cuda.current_stream(gt.device)
    key = (gt.device, stream.cuda_stream)
    with _WS_LOCK:
        ws = _WS.pop(key, None)           # (re-inserted below: the dict is in least-recently-used order)
        if ws is None or ws.numel() * 4 < nbytes:
            ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=gt.device)
        _WS[key] = ws
        while len(_WS) > _WS_MAX:         # callers that keep creating streams (CU-masked ExternalStreams, short-lived lanes) do not leak
            _WS.pop(next(iter(_WS)))      # one buffer per dead handle; an evicted buffer's block returns to the caching allocator
    st = C.c_void_p(stream.cuda_stream)
    _lib.check(lib.ivg_frame_metrics(C.c_void_p(gt.data_ptr()), dtype_code(gt.dtype), B, Tg, gt_t0, C.c_void_p(pred.data_ptr()), n, Tp, pred_t0,
                                     T, H, W, C.c_void_p(rows.data_ptr()), C.c_void_p(ws.data_ptr()), nbytes, st), None, "frame_metrics")
    return rows


class Evaluator:
    """``Evaluator(video_1, video_2)`` of the reference's eval loop (train_gpt.py:469-472; ivideogpt/utils/video_metric.py:63-100):
    video_1 = ground truth (B, T, 3, H, W), video_2 = predictions (t * B, T, 3, H, W), sample k of trajectory b at row k * B + b.
    -> ``(mse, psnr, ssim, lpips)`` scalars like the reference's 4-tuple: per-frame metrics, mean over a trajectory's frames, best of
    its t samples (min mse, max psnr / ssim), mean over trajectories -- computed by libivg ``ivg_frame_metrics``.  ``lpips`` is NaN:
    LPIPS (and FVD) need network weights that do not ship with the reference (out of scope); a caller that
    unpacks four values keeps working and sees an explicit not-a-number instead of a silently missing metric.
    ``rows(video_1, video_2)`` returns the per-trajectory (B, 3) rows -- the payload of the multi-GPU all-gather."""

    def __init__(self, i3d_path=None, max_batchsize=None):
        self.i3d_path, self.max_batchsize = i3d_path, max_batchsize     # accepted for signature compatibility; FVD is out of scope

    def rows(self, video_1, video_2):
        return frame_metric_rows(video_1, video_2)

    def __call__(self, video_1, video_2):
        m = self.rows(video_1, video_2).mean(0)
        return m[0], m[1], m[2], torch.full((), float("nan"), device=m.device)

    forward = __call__

    def eval(self):
        return self

    def to(self, *a, **k):
        return self
