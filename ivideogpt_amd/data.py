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
    "columbia_cairlab_pusht_real": (10, "image"),
    "stanford_kuka_multimodal_dataset_converted_externally_to_rlds": (20, "image"),
    "stanford_hydra_dataset_converted_externally_to_rlds": (10, "image"),
    "austin_buds_dataset_converted_externally_to_rlds": (20, "image"),
    "nyu_franka_play_dataset_converted_externally_to_rlds": (3, "image"),
    "maniskill_dataset_converted_externally_to_rlds": (20, "image"),
    "furniture_bench_dataset_converted_externally_to_rlds": (10, "image"),
    "ucsd_kitchen_dataset_converted_externally_to_rlds": (2, "image"),
    "ucsd_pick_and_place_dataset_converted_externally_to_rlds": (3, "image"),
    "austin_sailor_dataset_converted_externally_to_rlds": (20, "image"), "bc_z": (10, "image"),
    "utokyo_pr2_opening_fridge_converted_externally_to_rlds": (10, "image"),
    "utokyo_pr2_tabletop_manipulation_converted_externally_to_rlds": (10, "image"),
    "utokyo_xarm_pick_and_place_converted_externally_to_rlds": (10, "image"),
    "utokyo_xarm_bimanual_converted_externally_to_rlds": (10, "image"), "robo_net": (1, "image"),
    "kaist_nonprehensile_converted_externally_to_rlds": (10, "image"),
    "stanford_mask_vit_converted_externally_to_rlds": (1, "image"),
    "dlr_sara_pour_converted_externally_to_rlds": (10, "image"),
    "dlr_sara_grid_clamp_converted_externally_to_rlds": (10, "image"),
    "dlr_edan_shared_control_converted_externally_to_rlds": (5, "image"),
    "asu_table_top_converted_externally_to_rlds": (12.5, "image"),
    "iamlab_cmu_pickup_insert_converted_externally_to_rlds": (20, "image"), "uiuc_d3field1": (1, "image_1"),
    "uiuc_d3field2": (1, "image_2"), "uiuc_d3field3": (1, "image_3"), "uiuc_d3field4": (1, "image_4"),
    "utaustin_mutex": (20, "image"), "berkeley_fanuc_manipulation": (10, "image"), "cmu_playing_with_food": (10, "image"),
    "cmu_play_fusion": (5, "image"), "cmu_stretch": (10, "image"), "bair_robot_pushing": (1, "aux1_image"),
    "tfds_robonet": (1, "image"), "stanford_robocook_converted_externally_to_rlds1": (1, "image_1"),
    "stanford_robocook_converted_externally_to_rlds2": (1, "image_2"),
    "stanford_robocook_converted_externally_to_rlds3": (1, "image_3"),
    "stanford_robocook_converted_externally_to_rlds4": (1, "image_4"),
}
REFERENCE_STRIDE = DATASETS["fractal20220817_data"][0]


def resize_frames(images, size):
    """images float [T, 3, H, W] -> [T, 3, size, size], antialiased bilinear (== torchvision F.resize on tensors)."""
    if tuple(images.shape[-2:]) == (size, size):
        return images
    return F.interpolate(images, size=(size, size), mode="bilinear", antialias=True, align_corners=False)


def ingest_frames(frames_u8, size, center_crop=False, dtype=torch.float32):
    """uint8 frames [T, H, W, 3] ON THE GPU -> [T, 3, size, size] in [0, 1] (``dtype`` float32 / bfloat16): / 255, optional
    centre crop to the short side, antialiased bilinear resize -- one HIP kernel (libivg ``ivg_ingest_frames``) reading the
    interleaved rows as they are stored; the device-side form of ``resize_frames(frames / 255, size)``."""
    import ctypes as C
    from . import _lib
    from .packing import dtype_code
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
        raise AssertionError("ingest_frames: uint8 [T, H, W, 3] tensor on the GPU expected")
    src = frames_u8.contiguous()
    T, H, W, _ = src.shape
    out = torch.empty(T, 3, size, size, dtype=dtype, device=src.device)
    st = C.c_void_p(torch.cuda.current_stream(src.device).cuda_stream)
    _lib.check(_lib.load().ivg_ingest_frames(C.c_void_p(src.data_ptr()), T, H, W, int(bool(center_crop)), C.c_void_p(out.data_ptr()),
                                             dtype_code(dtype), int(size), st), None, "ingest_frames")
    return out


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

    def __init__(self, segment_length, image_size=64, device=None):
        """device=None: host preprocessing with the reference's exact arithmetic (what predict.py does before ``.to(device)``);
        device='cuda': the window's uint8 frames are uploaded as stored and preprocessed by the HIP ingest kernel."""
        self.segment_length, self.image_size, self.device = segment_length, image_size, device

    def parse(self, npz_file, dataset_name, load_action=False):
        episode = np.load(npz_file)
        rgb = episode[DATASETS.get(dataset_name, (1, "image"))[1]]
        window = pick_window(len(rgb), self.segment_length, frame_stride(dataset_name))
        if self.device is not None:
            frames = ingest_frames(torch.from_numpy(np.ascontiguousarray(rgb[window])).to(self.device), self.image_size)
            actions = torch.from_numpy(np.asarray(episode["action"][window])).float().to(self.device) if load_action else None
            return frames, actions      # both on self.device
        frames = torch.from_numpy(np.ascontiguousarray(rgb[window])).float().permute(0, 3, 1, 2)   # T,H,W,C -> T,C,H,W
        frames = resize_frames(frames / 255, self.image_size)
        actions = torch.from_numpy(np.asarray(episode["action"][window])).float() if load_action else None
        return frames, actions
