"""Clip ingest for the prediction path (host-side plumbing): the reference's per-episode ``.npz`` format and the
preprocessing of inference/utils.py:12-39.

  * episode file: key = display key (``image`` unless listed in DISPLAY_KEY), uint8 / int ``[T, H, W, 3]``; ``action`` float ``[T, A]``
    (datasets/oxe_data_converter.py:57-59, inference/samples/*.npz);
  * ``NPZParser.parse`` -> float ``[segment_length, 3, res, res]`` in [0, 1]: /255, then ``torchvision.transforms.functional
    .resize`` WITHOUT centre crop (aspect ratio is squashed, inference/utils.py:12-16).  For tensors torchvision's resize is
    ``torch.nn.functional.interpolate(mode='bilinear', antialias=True, align_corners=False)``, restated here (torchvision is
    not a dependency).  The segment start is drawn with ``np.random`` exactly like the reference (utils.py:23).
"""
import numpy as np
import torch
import torch.nn.functional as F

BASE_STEPSIZE = {'fractal20220817_data': 3, 'kuka': 10, 'bridge': 5, 'taco_play': 15, 'jaco_play': 10, 'berkeley_cable_routing': 10,
                 'roboturk': 10, 'viola': 20, 'toto': 30, 'language_table': 10, 'columbia_cairlab_pusht_real': 10,
                 'bair_robot_pushing': 1, 'tfds_robonet': 1, 'robo_net': 1, 'bc_z': 10, 'cmu_play_fusion': 5, 'cmu_stretch': 10}
DISPLAY_KEY = {'taco_play': 'rgb_static', 'roboturk': 'front_rgb', 'viola': 'agentview_rgb', 'language_table': 'rgb',
               'bair_robot_pushing': 'aux1_image', 'tfds_robonet': 'image'}


def resize_frames(images, size):
    """images float [T, 3, H, W] -> [T, 3, size, size], antialiased bilinear (== torchvision F.resize on tensors)."""
    if images.shape[-2] == size and images.shape[-1] == size:
        return images
    return F.interpolate(images, size=(size, size), mode="bilinear", antialias=True, align_corners=False)


class NPZParser:
    def __init__(self, segment_length, image_size=64):
        self.segment_length = segment_length
        self.image_size = image_size

    def preprocess(self, images):
        return resize_frames(images / 255, self.image_size)

    def get_segment(self, episode, actions, stepsize=1):
        if stepsize * self.segment_length > len(episode):   # shrink stepsize if the episode is too short
            stepsize = max(1, len(episode) // self.segment_length)
        start = np.random.randint(max(len(episode) - stepsize * self.segment_length + 1, 1))
        sl = slice(start, start + stepsize * self.segment_length, stepsize)
        return episode[sl], (actions[sl] if actions is not None else None)

    def get_stepsize(self, dataset_name):
        return max(round(BASE_STEPSIZE.get(dataset_name, 1) / BASE_STEPSIZE['fractal20220817_data']), 1)

    def parse(self, npz_file, dataset_name, load_action=False):
        data = np.load(npz_file)
        images = data[DISPLAY_KEY.get(dataset_name, 'image')]
        actions = data['action'] if load_action else None
        images, actions = self.get_segment(images, actions, self.get_stepsize(dataset_name))
        images = torch.from_numpy(np.ascontiguousarray(images)).float().permute(0, 3, 1, 2)   # T,H,W,C -> T,C,H,W
        images = self.preprocess(images)
        actions = torch.from_numpy(np.asarray(actions)).float() if actions is not None else None
        return images, actions
