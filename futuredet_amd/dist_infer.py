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


def pin_rank_cpus(local_rank, local_world):
    """Gives every rank of a node its own contiguous slice of the logical CPUs (e.g. 256 / 8 = 32 per rank) and sizes torch's
    intra-op pool to it: the host side of a rank is one launch thread + the pinned-memory copies, and 8 ranks that each start
    256 OpenMP / torch threads oversubscribe the box.  Returns the CPU ids (None when the platform has no affinity call)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    per = max(1, len(cpus) // local_world)
    mine = cpus[local_rank * per:(local_rank + 1) * per] or cpus
    os.sched_setaffinity(0, mine)
    torch.set_num_threads(max(1, min(len(mine), 8)))
    return mine


def sync_replicas(model, src=0, check=True):
    """What the reference gets from wrapping the model in DistributedDataParallel (tools/dist_test.py:177-188: the DDP
    constructor broadcasts rank 0's parameters and buffers): every rank ends up with rank ``src``'s weights, in ONE
    broadcast of a flattened copy, and -- ``check`` -- an all_gather of a float64 checksum proves the replicas agree
    (raises otherwise).  Single process: no-op.  Returns the checksum."""
    tensors = [t for t in list(model.parameters()) + list(model.buffers()) if t.is_floating_point() or t.dtype in (torch.int64, torch.int32)]
    with torch.no_grad():
        def checksum():
            return float(sum(t.detach().double().sum() + 3.0 * t.detach().double().abs().sum() for t in tensors)) if tensors else 0.0

        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return checksum()
        dev = tensors[0].device if dist.get_backend() == "nccl" else torch.device("cpu")
        flat = torch.cat([t.detach().reshape(-1).to(device=dev, dtype=torch.float64) for t in tensors])
        dist.broadcast(flat, src=src)
        o = 0
        for t in tensors:
            n = t.numel()
            t.copy_(flat[o:o + n].reshape(t.shape).to(device=t.device, dtype=t.dtype))  # in-place: bumps _version, derived caches follow
            o += n
        cs = checksum()
        if check:
            mine = torch.tensor([cs], dtype=torch.float64, device=dev)
            allc = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
            dist.all_gather(allc, mine)
            vals = [float(v) for v in allc]
            if any(v != vals[0] for v in vals):
                raise RuntimeError("model replicas differ after the broadcast: checksums %s" % vals)
        return cs


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
