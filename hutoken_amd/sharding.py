"""Multi-GPU sharding of a packed batch: documents are independent, so a batch is cut
into contiguous document ranges (one per rank) and the only exchange is an all-gather
of the per-rank id totals, from which every rank knows where its ids sit in the global
id stream.  Works over any torch.distributed backend ("nccl" = RCCL on the GPUs,
"gloo" in the CPU tests)."""
import numpy as np


def shard_by_docs(n_docs, world):
    """Equal document counts: [(first, count)] * world."""
    base, extra = divmod(n_docs, world)
    out, at = [], 0
    for r in range(world):
        c = base + (1 if r < extra else 0)
        out.append((at, c))
        at += c
    return out


def shard_by_bytes(offsets, world):
    """Contiguous document ranges of about equal BYTES (prefix sums of the lengths):
    [(first, count)] * world."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n = len(offsets) - 1
    total = int(offsets[n])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(offsets, total * r // world, side="left")))
    cuts.append(n)
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world)]


def local_view(data, offsets, first, count):
    """Bytes and rebased offsets of documents [first, first+count)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    a, b = int(offsets[first]), int(offsets[first + count])
    return data[a:b], offsets[first:first + count + 1] - a


def gather_id_totals(local_total, device=None):
    """All-gather of one int64 per rank -> list of totals (length world)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    t = torch.tensor([int(local_total)], dtype=torch.int64, device=device)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [int(x.item()) for x in out]


def global_id_base(totals, rank):
    """Position of this rank's first id in the global id stream."""
    return int(sum(totals[:rank]))
