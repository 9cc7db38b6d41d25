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


def _parse_cpulist(text):
    """'0-31,128-159' -> [0..31, 128..159]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def gpu_numa_node(index, sysfs="/sys"):
    """NUMA node of GPU ``index`` from its PCI function's sysfs entry (None when unknown: no such device, the kernel reports -1, or
    the torch build does not expose the PCI address)."""
    try:
        pr = torch.cuda.get_device_properties(index)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        node = int(open(os.path.join(sysfs, "bus/pci/devices", bdf, "numa_node")).read())
        return node if node >= 0 else None
    except Exception:
        return None


def rank_cpu_slices(cpus, local_world, gpu_nodes=None, node_cpus=None):
    """CPU ids per local rank.  With the GPUs' NUMA nodes known (``gpu_nodes[r]``) and the nodes' CPU lists (``node_cpus[n]``), the ranks
    whose GPUs hang off one node share THAT node's allowed CPUs in equal contiguous slices (the launch thread and the pinned result buffers
    stay next to the GPU's root complex); a rank whose node is unknown, or whose node has no allowed CPU, takes its contiguous slice of all
    CPUs -- the rule used when nothing is known.  Pure function (tested on CPU with made-up topologies)."""
    cpus = sorted(cpus)
    per = max(1, len(cpus) // local_world)
    out = [cpus[r * per:(r + 1) * per] or cpus for r in range(local_world)]
    if not gpu_nodes or not node_cpus:
        return out
    allowed = set(cpus)
    for node in sorted({n for n in gpu_nodes if n is not None}):
        mine = [c for c in node_cpus.get(node, []) if c in allowed]
        ranks = [r for r in range(local_world) if gpu_nodes[r] == node]
        if not mine or len(mine) < len(ranks):
            continue
        k = len(mine) // len(ranks)
        for j, r in enumerate(ranks):
            out[r] = mine[j * k:(j + 1) * k]
    return out


def pin_rank_cpus(local_rank, local_world, sysfs="/sys"):
    """Gives every rank of a node its own slice of the logical CPUs (e.g. 256 / 8 = 32 per rank), taken from the NUMA node its GPU is
    attached to when sysfs tells (rank_cpu_slices), and sizes torch's intra-op pool to it: the host side of a rank is one launch thread +
    the pinned-memory copies, and 8 ranks that each start 256 OpenMP / torch threads oversubscribe the box.  Returns the CPU ids (None
    when the platform has no affinity call)."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    cpus = sorted(os.sched_getaffinity(0))
    gpu_nodes, node_cpus = None, None
    if torch.cuda.is_available() and torch.cuda.device_count() >= local_world:
        gpu_nodes = [gpu_numa_node(r, sysfs) for r in range(local_world)]
        node_cpus = {}
        for n in {g for g in gpu_nodes if g is not None}:
            try:
                node_cpus[n] = _parse_cpulist(open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % n)).read())
            except Exception:
                pass
    mine = rank_cpu_slices(cpus, local_world, gpu_nodes, node_cpus)[local_rank]
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
    if bool((counts < 0).any()):  # fd_decode_cfg, nms_kind 1: a group the circular NMS could not decide from the candidates it took
        raise RuntimeError("a decode group reports count -1 (circular NMS, more candidates than the kernels take: see include/futuredet_hip.h)")
    valid = torch.arange(post, device=packed.device).view(1, 1, post) < counts.unsqueeze(-1).to(packed.device)
    out = []
    for i in range(n):
        m = valid[i].reshape(-1)
        flat = packed[i].reshape(-1, 11)[m]
        out.append({"box3d_lidar": flat[:, :9], "scores": flat[:, 9], "label_preds": flat[:, 10].long()})
    return out
