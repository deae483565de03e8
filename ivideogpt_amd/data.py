"""Clip ingest for the prediction path (host-side plumbing): the reference's per-episode ``.npz`` format and the
preprocessing of inference/utils.py:12-39.

  * episode file: one uint8 / int ``[T, H, W, 3]`` array under the dataset's display key (``image`` unless the table below
    names another) and a float ``[T, A]`` array ``action`` (datasets/oxe_data_converter.py:57-59, inference/samples/*.npz);
  * ``NPZParser.parse`` -> float ``[segment_length, 3, res, res]`` in [0, 1]: a strided window of the episode, /255, then
    torchvision's tensor ``resize`` WITHOUT centre crop (the aspect ratio is squashed, inference/utils.py:12-16) -- for tensors
    that is ``interpolate(mode='bilinear', antialias=True, align_corners=False)``, restated here (torchvision is not a
    dependency).  The window start is drawn from ``np.random.randint`` with the reference's bound (utils.py:23), so a seeded
    run picks the same frames.
"""
import numpy as np
import torch
import torch.nn.functional as F

# dataset -> (frame stride at the dataset's native rate, key of the RGB stream in the episode file); everything is resampled
# relative to fractal20220817_data (3 Hz-equivalent stride), as the reference's table does
DATASETS = {
    "fractal20220817_data": (3, "image"), "kuka": (10, "image"), "bridge": (5, "image"), "taco_play": (15, "rgb_static"),
    "jaco_play": (10, "image"), "berkeley_cable_routing": (10, "image"), "roboturk": (10, "front_rgb"),
    "viola": (20, "agentview_rgb"), "toto": (30, "image"), "language_table": (10, "rgb"),
    "columbia_cairlab_pusht_real": (10, "image"), "bair_robot_pushing": (1, "aux1_image"), "tfds_robonet": (1, "image"),
    "robo_net": (1, "image"), "bc_z": (10, "image"), "cmu_play_fusion": (5, "image"), "cmu_stretch": (10, "image"),
}
REFERENCE_STRIDE = DATASETS["fractal20220817_data"][0]


def resize_frames(images, size):
    """images float [T, 3, H, W] -> [T, 3, size, size], antialiased bilinear (== torchvision F.resize on tensors)."""
    if tuple(images.shape[-2:]) == (size, size):
        return images
    return F.interpolate(images, size=(size, size), mode="bilinear", antialias=True, align_corners=False)


def frame_stride(dataset_name):
    native = DATASETS.get(dataset_name, (1, "image"))[0]
    return max(1, round(native / REFERENCE_STRIDE))


def pick_window(n_frames, length, stride):
    """-> slice of `length` frames every `stride`; the stride shrinks when the episode is too short, the start is random."""
    if stride * length > n_frames:
        stride = max(1, n_frames // length)
    span = stride * length
    first = np.random.randint(max(n_frames - span + 1, 1))
    return slice(first, first + span, stride)


class NPZParser:
    """Same constructor / ``parse`` contract as the reference's parser (inference/utils.py:18-39)."""

    def __init__(self, segment_length, image_size=64):
        self.segment_length, self.image_size = segment_length, image_size

    def parse(self, npz_file, dataset_name, load_action=False):
        episode = np.load(npz_file)
        rgb = episode[DATASETS.get(dataset_name, (1, "image"))[1]]
        window = pick_window(len(rgb), self.segment_length, frame_stride(dataset_name))
        frames = torch.from_numpy(np.ascontiguousarray(rgb[window])).float().permute(0, 3, 1, 2)   # T,H,W,C -> T,C,H,W
        frames = resize_frames(frames / 255, self.image_size)
        actions = torch.from_numpy(np.asarray(episode["action"][window])).float() if load_action else None
        return frames, actions
