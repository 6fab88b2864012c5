"""Frame-parallel Monte-Carlo plumbing shared by bench.py and the FER tools.

Frames are independent, so a batch shards contiguously over ranks with no data-path collective; the only
exchange is the final sum of two counters (block errors, bit errors) -- RCCL over xGMI on GPUs
(torch.distributed backend "nccl"), gloo in the CPU tests.  The reference's main() does the same
accounting sequentially (CASCL_1024_L8.c:296-305)."""
import torch


def frame_shard(total, rank, world):
    """Contiguous shard [start, start+count) of `total` frames for `rank` of `world` (sizes differ by <= 1)."""
    base, rem = divmod(total, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def allreduce_counters(counters, dist=None):
    """counters: int64 tensor [2] = (block errors, bit errors) of this rank; summed over ranks in place."""
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counters, op=dist.ReduceOp.SUM)
    return counters


def max_over_ranks(seconds, device, dist=None):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sequential_stop_cut(frame_err, ble):
    """The reference stops a SNR point at the frame where the `ble`-th block error occurs
    (`for (run = 0; errBlock < BLE; run++)`, SCL_1024.c:228).  Given per-frame bit-error counts of a batch
    decoded in frame order, return (run, block_errors, bit_errors) at that cut, or None if the batch does not
    reach `ble` errors."""
    fe = torch.as_tensor(frame_err).to(torch.int64).flatten()
    bad = (fe > 0).to(torch.int64)
    cum = torch.cumsum(bad, 0)
    hit = torch.nonzero(cum >= ble)
    if hit.numel() == 0:
        return None
    run = int(hit[0].item()) + 1
    return run, ble, int(fe[:run].sum().item())


def fer_point(decode_and_count, total_frames, rank=0, world=1, dist=None, device="cpu"):
    """One SNR point: every rank decodes its shard with `decode_and_count(start, count) -> (blk, bits)` and the
    two counters are summed over ranks.  Returns (block_errors, bit_errors, frames)."""
    start, count = frame_shard(total_frames, rank, world)
    blk, bits = decode_and_count(start, count)
    c = torch.tensor([int(blk), int(bits)], dtype=torch.int64, device=device)
    allreduce_counters(c, dist)
    return int(c[0].item()), int(c[1].item()), total_frames


def gather_frame_errors(frame_err_local, total_frames, rank=0, world=1, dist=None):
    """Per-frame bit-error counts of this rank's contiguous shard (frame_shard order) -> the counts of all
    `total_frames` frames in frame order on every rank.  This is the one extra exchange the exact sequential stop
    rule needs when a batch is spread over GPUs (4 bytes per frame); shards differ by at most one frame,
    so they are padded to a common length for all_gather."""
    fe = torch.as_tensor(frame_err_local).to(torch.int32).flatten()
    if dist is None or not dist.is_initialized() or world == 1:
        return fe
    base, rem = divmod(total_frames, world)
    width = base + (1 if rem else 0)
    pad = torch.zeros(width, dtype=torch.int32, device=fe.device)
    pad[:fe.numel()] = fe
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = []
    for r in range(world):
        _, cnt = frame_shard(total_frames, r, world)
        out.append(parts[r][:cnt])
    return torch.cat(out)


def sequential_stop_cut_sharded(frame_err_local, total_frames, ble, rank=0, world=1, dist=None):
    """sequential_stop_cut over a batch that was decoded in shards: same (run, block_errors, bit_errors) on every
    rank as one process decoding the whole batch would get."""
    return sequential_stop_cut(gather_frame_errors(frame_err_local, total_frames, rank, world, dist), ble)
