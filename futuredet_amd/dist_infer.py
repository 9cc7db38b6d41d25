"""Multi-GPU inference: one process per GPU, samples sharded rank-strided, one fixed-shape all_gather.

Mirrors tools/dist_test.py:125-135,169-188,236-252 of the reference (init_process_group("nccl", "env://"),
DistributedSampler(shuffle=False) = rank-strided sample split, barrier + all_gather of the per-rank results),
with the pickled-object gather (torchie/trainer/utils.py:115-155) replaced by one all_gather of fixed-shape
tensors [n_local, S, post, 11] (box 9 + score + label) and [n_local, S] counts.  On ROCm backend "nccl" is RCCL
(xGMI inside a node); the same code runs over "gloo" on CPU tensors in the tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, init_method="env://", rank=rank, world_size=world)
    return rank, world, local


def get_dist_info():
    """det3d/torchie/trainer/utils.py:22-34 -> (rank, world_size)"""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def synchronize():
    """det3d/torchie/trainer/utils.py:100-112: barrier among all ranks (no-op single process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def all_gather(data):
    """det3d/torchie/trainer/utils.py:115-155 contract (any picklable object in, list with one entry per rank out), as
    tools/dist_test.py:237 uses it on the token -> detections dict.  Convenience for callers written against the
    reference; the benchmarked path gathers fixed-shape tensors instead (gather_results)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [data]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, data)
    return out


def shard_indices(n_samples, rank, world):
    """DistributedSampler(shuffle=False) rule (det3d/datasets/loader/build_loader.py:38): rank r takes r, r+W, ...;
    the tail is padded by wrapping so every rank has the same count."""
    per = (n_samples + world - 1) // world
    idx = list(range(n_samples))
    idx += idx[: per * world - n_samples]
    return idx[rank: per * world: world]


def pack_results(boxes, scores, labels, counts):
    """[n,S,post,9],[n,S,post],[n,S,post] int64,[n,S] -> ([n,S,post,11] float32, [n,S] int32)"""
    packed = torch.cat([boxes.float(), scores.float().unsqueeze(-1), labels.float().unsqueeze(-1)], dim=-1)
    return packed.contiguous(), counts.int().contiguous()


def gather_results(packed, counts, n_samples=None):
    """All ranks receive every rank's results, re-interleaved into global sample order (inverse of
    shard_indices).  Single-process: identity."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return packed, counts
    world = dist.get_world_size()
    outs = [torch.empty_like(packed) for _ in range(world)]
    cnts = [torch.empty_like(counts) for _ in range(world)]
    dist.all_gather(outs, packed)
    dist.all_gather(cnts, counts)
    full = torch.stack(outs, dim=1).reshape(-1, *packed.shape[1:])   # sample i*W + r  <- rank r, local i
    fullc = torch.stack(cnts, dim=1).reshape(-1, *counts.shape[1:])
    if n_samples is not None:
        full, fullc = full[:n_samples], fullc[:n_samples]
    return full, fullc


def unpack_results(packed, counts):
    """-> list (per sample) of dicts with box3d_lidar [K,9], scores [K], label_preds [K] (reference output keys)."""
    n, S, post, _ = packed.shape
    valid = torch.arange(post, device=packed.device).view(1, 1, post) < counts.unsqueeze(-1).to(packed.device)
    out = []
    for i in range(n):
        m = valid[i].reshape(-1)
        flat = packed[i].reshape(-1, 11)[m]
        out.append({"box3d_lidar": flat[:, :9], "scores": flat[:, 9], "label_preds": flat[:, 10].long()})
    return out
