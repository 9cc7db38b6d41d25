"""Batch assembly for the path: collate_kitti_multi's hot keys (det3d/torchie/parallel/collate.py:163-245),
example_to_device (det3d/torchie/apis/train.py:28-71) and the inference branch of batch_processor (:106-126)."""
import collections

import numpy as np
import torch


def collate_kitti_multi(batch_list, samples_per_gpu=1):
    merged = collections.defaultdict(list)
    for example in batch_list:
        for sub in (example if isinstance(example, list) else [example]):
            for k, v in sub.items():
                merged[k].append(v)
    ret = {}
    for key, elems in merged.items():
        if key in ("voxels", "num_points", "num_voxels"):
            ret[key] = torch.tensor(np.concatenate(elems, axis=0))
        elif key == "metadata":
            ret[key] = elems
        elif key in ("coordinates", "points"):
            # collate.py:199-206: prefix the sample index as a leading column
            ret[key] = torch.tensor(np.concatenate(
                [np.pad(c, ((0, 0), (1, 0)), mode="constant", constant_values=i) for i, c in enumerate(elems)], axis=0))
        elif key == "bev_map":
            ret[key] = [torch.tensor(np.stack([e[i] for e in elems], axis=0)) for i in range(len(elems[0]))]
        else:
            ret[key] = np.stack(elems, axis=0)
    return ret


def example_to_device(example, device, non_blocking=False):
    out = {}
    for k, v in example.items():
        if k in ("voxels", "bev_map", "coordinates", "num_points", "points", "num_voxels"):
            if isinstance(v, list):
                out[k] = [t.to(device, non_blocking=non_blocking) for t in v]
            else:
                out[k] = v.to(device, non_blocking=non_blocking)
        else:
            out[k] = v
    return out


def batch_processor(model, data, train_mode, **kwargs):
    """det3d/torchie/apis/train.py:106-126, the call tools/dist_test.py:177 makes per batch: move the collated example to
    the rank's device (``local_rank`` keyword) and run the detector.  Only the inference branch exists on this path."""
    if train_mode:
        raise NotImplementedError("training (losses, optimiser) is outside the inference hot path (SURVEY 2)")
    device = torch.device("cuda", kwargs["local_rank"]) if "local_rank" in kwargs else None
    assert device is not None, "batch_processor needs local_rank (the reference asserts the same in example_to_device)"
    example = example_to_device(data, device, non_blocking=False)
    del data
    return model(example, return_loss=False)
