"""Frame-range sharding of one chunk across the GPUs of a node (one process per GPU).

The reference is a single process; a chunk's frames are independent
except for three things, which are the only exchanges made here (all small, all over
torch.distributed -- RCCL over xGMI when the backend is "nccl"):

1. the background window (ip.py:189-190) and the flow of a shard's first frame (ip.py:856) read
   up to ``radius`` frames of the neighbouring shards: every rank LOADS those halo frames with its
   own range (no collective: they come from the decoder);
2. persistence (ip.py:270-289) is a run-length over the whole chunk: ranks all-gather the first and
   last T-1 raw planes of every shard, evaluate their own planes extended by the neighbours' edge
   planes, and OR the resulting maps (all-gather of one plane per rank + ipx_or_planes);
3. the static map (ip.py:438-465) needs chunk-wide heat / valid sums: one all-reduce(sum) of two
   int32 H x W maps; every rank then thresholds identically.

Per-frame results (5 + 3 integers, flow, keep) are all-gathered at the end.
"""
import numpy as np
import torch

from . import hostmath


def shard_range(n_frames, rank, world):
    """Frames [a, b) of an n-frame chunk owned by ``rank``."""
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def halo_range(n_frames, a, b, radius):
    """Frames [lo, hi) a rank must hold to process [a, b): its own plus ``radius`` (>= 1: the flow of
    frame a reads frame a-1) frames each side, clipped at the chunk edges (the clip is part of the
    reference's semantics, ip.py:189-190)."""
    r = max(int(radius), 1)
    return max(0, a - r), min(n_frames, b + r)


class _Solo:
    """Collectives of a world of one (no process group needed)."""
    world = 1
    rank = 0
    exchange = False                                     # no rank to exchange with: the collective steps are skipped

    def all_gather(self, t):
        return [t]

    def all_reduce_sum(self, t):
        return t


class _Dist:
    def __init__(self, group=None, forced=False):
        import torch.distributed as dist
        self.dist = dist
        self.calls = 0                                   # collectives issued (bench.py reports them per chunk)
        self.bytes = 0                                   # payload bytes this rank contributed
        self.timed = False                               # set: every collective is bracketed by events on its stream
        self._events = []
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # RCCL ("nccl") moves device tensors directly; gloo (CPU rehearsals of the multi-rank path) is given
        # host copies, because it has no device all_gather
        self.stage = dist.get_backend(group) == "gloo"
        # a world of one normally skips the exchanges; `forced` runs every collective all the same
        self.exchange = self.world > 1 or bool(forced)

    def _bracket(self, t):
        if not (self.timed and t.is_cuda):
            return None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._events.append((e0, e1))
        return e1

    def collective_ms(self, reset=True):
        """GPU time between the start and the end of every timed collective since the last reset (synchronises)."""
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._events)
        if reset:
            self._events, self.calls, self.bytes = [], 0, 0
        return ms

    def all_gather(self, t):
        self.calls += 1
        self.bytes += t.numel() * t.element_size()
        end = self._bracket(t)
        if self.stage and t.is_cuda:
            h = t.contiguous().cpu()
            out = [torch.empty_like(h) for _ in range(self.world)]
            self.dist.all_gather(out, h, group=self.group)
            out = [o.to(t.device) for o in out]
        else:
            out = [torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(out, t.contiguous(), group=self.group)
        if end is not None:
            end.record()
        return out

    def all_reduce_sum(self, t):
        self.calls += 1
        self.bytes += t.numel() * t.element_size()
        end = self._bracket(t)
        if self.stage and t.is_cuda:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        if end is not None:
            end.record()
        return t


def make_comm(group=None, force=False):
    """The collectives of ``group``: torch.distributed when a process group of more than one rank is up, else the
    world-of-one stand-in.  ``force`` takes the torch.distributed path for a world of one as well (how the RCCL
    calls of this module are exercised on a one-GPU box)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and (force or dist.get_world_size(group) > 1):
        return _Dist(group, forced=force)
    return _Solo()


def detect_chunk_sharded(engine, ext_frames, a, b, n_frames, comm=None, want_final_bits=False):
    """Run the detection path for frames [a, b) of an ``n_frames`` chunk on this rank.

    engine      : Engine (or an object with the same stage methods; tests inject a CPU stand-in)
    ext_frames  : (hi-lo, H, W) uint8 tensor holding chunk frames halo_range(n_frames, a, b, radius)
    Returns a dict of GLOBAL per-frame tensors (identical on every rank): flows, keep, counts (n,5),
    cluster (n,3); plus this rank's ``final_bits`` for its own frames on request.
    """
    comm = comm or make_comm()
    exchange = getattr(comm, "exchange", comm.world > 1)
    radius = int(engine.params.sliding_window_radius)
    T = int(engine.params.consecutive_threshold)
    lo, hi = halo_range(n_frames, a, b, radius)
    assert ext_frames.shape[0] == hi - lo, "ext_frames must hold frames [%d, %d)" % (lo, hi)
    _, H, W = ext_frames.shape
    hl, nown = a - lo, b - a
    if nown <= 0:
        raise ValueError("every rank needs at least one frame of the chunk")
    edge = max(T - 1, 0)
    if exchange and nown < edge:
        raise ValueError("frame-range sharding needs at least consecutive_threshold-1 = %d frames per rank" % edge)
    own = ext_frames[hl:hl + nown]

    # motion test for the owned frames (frame a-1 is the last left-halo frame)
    flows, keep = engine.flow_metric(own, prev_frame=ext_frames[hl - 1] if a > 0 else None)
    keep_ext = torch.ones(hi - lo, dtype=torch.uint8, device=ext_frames.device)
    keep_ext[hl:hl + nown] = keep

    # raw planes of the owned frames, written into the middle of [edge | own | edge]
    stack = engine.empty_bits(nown + 2 * edge, H, W)
    raw = stack[edge:edge + nown]
    _, bright, dense = engine.detect_raw(ext_frames, keep=keep_ext, first=hl, count=nown, raw_out=raw)
    engine.remove_bright_regions(bright, raw, H, W, out=raw)

    # persistence across shard boundaries
    if exchange and edge > 0:
        mine = torch.cat([raw[:edge], raw[nown - edge:]], dim=0).contiguous()
        edges = comm.all_gather(mine)
        stack[:edge] = edges[comm.rank - 1][edge:] if comm.rank > 0 else 0
        stack[edge + nown:] = edges[comm.rank + 1][:edge] if comm.rank + 1 < comm.world else 0
        pers = engine.persistent_mask(stack, None, H, W)
        pers = engine.or_planes(torch.stack(comm.all_gather(pers)), H, W)
    elif exchange:
        pers = engine.persistent_mask(raw, keep, H, W)         # T <= 1: nothing crosses a boundary
        pers = engine.or_planes(torch.stack(comm.all_gather(pers)), H, W)
    else:
        pers = engine.persistent_mask(raw, keep, H, W)

    # static map from chunk-wide heat / valid sums
    maps = engine.static_maps(own, raw, pers, keep)
    if exchange:
        maps = comm.all_reduce_sum(maps)
    static = engine.static_from_maps(maps)

    fin, c3 = engine.finalize(own, raw, pers, static, keep, want_bits=True)
    cluster, _ = engine.cluster_stats(own, fin, keep)

    # gather the per-frame scalars (padded to the largest shard)
    if exchange:
        sizes = [shard_range(n_frames, r, comm.world) for r in range(comm.world)]
        mx = max(e - s for s, e in sizes)
        pack = torch.zeros((mx, 10), dtype=torch.int64, device=ext_frames.device)
        pack[:nown, 0:3] = c3
        pack[:nown, 3:5] = dense
        pack[:nown, 5:8] = cluster
        pack[:nown, 8] = keep.to(torch.int64)
        pack[:nown, 9] = flows.view(torch.int32).to(torch.int64)
        parts = comm.all_gather(pack)
        allp = torch.cat([parts[r][:e - s] for r, (s, e) in enumerate(sizes)], dim=0)
    else:
        allp = torch.cat([c3, dense, cluster, keep.to(torch.int64)[:, None],
                          flows.view(torch.int32).to(torch.int64)[:, None]], dim=1)
    out = dict(counts=allp[:, 0:5].contiguous(), cluster=allp[:, 5:8].contiguous(), keep=allp[:, 8].to(torch.uint8),
               flows=allp[:, 9].to(torch.int32).view(torch.float32))
    if want_final_bits:
        out["final_bits"] = fin
    return out


def results_to_host(out, n_pixels):
    """Device tensors of detect_chunk_sharded -> the four float64 arrays of ip.py:632 (+ flows, keep)."""
    keep = out["keep"].cpu().numpy()
    total, cc, cs, cb = hostmath.chunk_results(keep, out["counts"].cpu().numpy(), out["cluster"].cpu().numpy(), n_pixels)
    return dict(total_counts=total, cluster_counts=cc, avg_sizes=cs, avg_brightnesses=cb,
                flows=out["flows"].cpu().numpy().astype(np.float64), keep=keep)
