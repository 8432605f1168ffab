"""Frame sharding and the result gather of the multi-GPU path.

Frames are independent (reference: `detect(&self)` only reads immutable fields,
src/detector.rs:17-23,505), so a batch shards by frame with no data-path collective; the one
exchange step is the gather of the per-frame results to rank 0.  Works on any torch.distributed
backend: "nccl" (= RCCL over xGMI) with device tensors in bench.py, "gloo" with CPU tensors in
the tests.
"""
import torch
import torch.distributed as dist

SLAB_RECORDS = 512  # average saddle records per frame the packed result buffer holds (the chain
# packs the frames' lists back to back; a batch that needs more is flagged, never truncated)


def shard_range(rank, world, frames_per_rank):
    """Global frame indices owned by `rank` (weak scaling: every rank owns frames_per_rank)."""
    lo = rank * frames_per_rank
    return lo, lo + frames_per_rank


def alloc_packed(n_frames, device, slab_records=SLAB_RECORDS):
    """One rank's result slab as ONE device buffer: [n_frames * 4 table words | n_frames * slab_records * 5 record
    floats] (float32 storage; the table is an int32 view of its head).  -> (flat, saddles, table): the two views are the
    caller-owned buffers of agx_saddles_batch_enqueue_to, `flat` is what the gather moves -- one message per rank and
    step instead of two."""
    flat = torch.zeros(n_frames * 4 + n_frames * slab_records * 5, dtype=torch.float32, device=device)
    return (flat,) + split_packed(flat, n_frames)


def split_packed(flat, n_frames):
    """(saddles [n, 5] f32, table [n_frames, 4] i32) views of a packed slab."""
    return flat[n_frames * 4:].view(-1, 5), flat[: n_frames * 4].view(torch.int32).view(n_frames, 4)


def alloc_result_buffers(n_frames, device, slab_records=SLAB_RECORDS):
    """(saddles [n_frames*slab_records, 5] f32, table [n_frames, 4] i32: count, offset, status,
    clusters) -- the caller-owned device buffers of agx_saddles_batch_enqueue_to (views of one packed slab)."""
    return alloc_packed(n_frames, device, slab_records)[1:]


def gather_results(saddles, table, dst=0, group=None):
    """Gather every rank's (saddles, table) to `dst`.  Returns (list_of_saddles, list_of_tables)
    on dst and (None, None) elsewhere.  world_size 1: no communication."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [saddles], [table]
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if rank == dst:
        gs = [torch.empty_like(saddles) for _ in range(world)]
        gt = [torch.empty_like(table) for _ in range(world)]
        dist.gather(table, gt, dst=dst, group=group)
        dist.gather(saddles, gs, dst=dst, group=group)
        return gs, gt
    dist.gather(table, None, dst=dst, group=group)
    dist.gather(saddles, None, dst=dst, group=group)
    return None, None


class GatherPipeline:
    """Result buffers with asynchronous gathers, so that the gather of a step overlaps the chain of the
    following ones (which write other buffers).  A rank's frame table and its records live in ONE packed
    buffer (alloc_packed), so a gather is one message per rank: on rank 0 of an 8-GPU node 7 receives
    instead of 14 (tables and records apart).

    `steps_per_gather` = k > 1: the slabs of k consecutive steps are one contiguous buffer and go to rank
    `dst` in ONE collective when the k-th is enqueued (every step's results are delivered, the first of a
    group k - 1 steps late; finish() sends a group that is not full yet) -- fewer, larger messages: the cost
    of a collective beside the chain is mostly per collective, not per byte.
    `every` > 1 (with k = 1) gathers only every n-th submitted step and always the last one before finish():
    what a consumer that polls results at a lower rate than the chain produces them would ask for.

        pipe = GatherPipeline(n_frames, device)
        for step in ...:
            out, table = pipe.acquire()      # waits (stream-side) for the gather that last used them
            det.saddles_batch_enqueue_to(frames, out, table)
            pipe.submit()                    # async gather of the packed slab(s) to rank `dst`
        gathered = pipe.finish()             # on dst: (list_of_saddles, list_of_tables) of the LAST step
    """

    def __init__(self, n_frames, device, dst=0, group=None, depth=2, always_depth=False, slab_records=SLAB_RECORDS,
                 force_collective=False, every=1, steps_per_gather=1):
        # force_collective: a world of ONE rank still sends its slabs through the backend's gather (bench.py
        # --collective-world-1: what a one-GPU box can exercise of the nccl = RCCL path)
        self.dst, self.group = dst, group
        self.n_frames = n_frames
        self.multi = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force_collective)
        self.world = dist.get_world_size(group) if self.multi else 1
        self.rank = dist.get_rank(group) if self.multi else 0
        self.every = max(1, int(every))
        self.k = max(1, int(steps_per_gather)) if self.multi else 1
        assert self.every == 1 or self.k == 1, "gather every n-th step OR several steps per gather"
        # one rank alone needs a single buffer unless several batches are in flight (ChainPipeline)
        n_groups = depth if (self.multi or always_depth) else 1
        slab = n_frames * 4 + n_frames * slab_records * 5  # floats of one packed slab
        self.slab = slab
        # a group = k slabs in one allocation (what one collective moves); bufs = the slabs' views in the order they are used
        self.group_flat = [torch.zeros(self.k * slab, dtype=torch.float32, device=device) for _ in range(n_groups)]
        self.flat = [g[j * slab:(j + 1) * slab] for g in self.group_flat for j in range(self.k)]
        self.bufs = [split_packed(f, n_frames) for f in self.flat]
        self.recv = None
        if self.multi and self.rank == dst:
            self.recv = [[torch.empty_like(g) for _ in range(self.world)] for g in self.group_flat]
        self.works = [None] * n_groups
        self.i = -1
        self.submitted = 0
        self.gathered_i = None  # slot index of the last step that went through a gather

    def acquire(self):
        self.i = (self.i + 1) % len(self.bufs)
        if self.i % self.k == 0:  # the first slab of a group: the gather that last read the group must be through
            g = self.i // self.k
            w = self.works[g]
            if w is not None:
                w.wait()  # the current stream waits for it
                self.works[g] = None
        return self.bufs[self.i]

    def _gather(self):
        g = self.i // self.k
        self.works[g] = dist.gather(self.group_flat[g], self.recv[g] if self.rank == self.dst else None, dst=self.dst,
                                    group=self.group, async_op=True)
        self.gathered_i = self.i

    def submit(self):
        self.submitted += 1
        if not self.multi:
            return
        if self.k > 1:
            if self.i % self.k == self.k - 1:  # the group is full
                self._gather()
        elif self.submitted % self.every == 0:
            self._gather()

    def finish(self):
        if self.multi and self.gathered_i != self.i and self.i >= 0:
            self._gather()  # (a group that is not full yet / every > 1: the last step's results are always delivered)
        for w in self.works:
            if w is not None:
                w.wait()
        self.works = [None] * len(self.works)
        if not self.multi:
            return [self.bufs[self.i][0]], [self.bufs[self.i][1]]
        if self.rank != self.dst:
            return None, None
        g, j = self.i // self.k, self.i % self.k
        views = [split_packed(r[j * self.slab:(j + 1) * self.slab], self.n_frames) for r in self.recv[g]]
        return [v[0] for v in views], [v[1] for v in views]


class ChainPipeline:
    """Several batches in flight on one GPU: `depth` detectors, each with its own workspace and HIP
    stream, take the batches in turn, so that the dense kernel (K1) of batch i+1 runs while the
    sparse kernels (verify .. filter) of batch i -- short, latency-bound launches that leave most
    of the chip idle -- are still going.  Results go to `depth` device buffer pairs and, with
    several ranks, through the asynchronous gather of GatherPipeline.

        pipe = ChainPipeline(tag_family, n_frames, device, depth=2)
        for frames in batches:            # [n_frames, H, W] uint8 device tensors
            pipe.submit(frames)           # ordered behind the current stream's work on `frames`
        saddles, tables = pipe.finish()   # results of the LAST batch (lists over ranks on dst)
    """

    def __init__(self, tag_family, n_frames, device, depth=2, params=None, dst=0, group=None, slab_records=SLAB_RECORDS,
                 detector_cls=None, force_collective=False, gather_every=1, steps_per_gather=1):
        # detector_cls: a stand-in with TagDetector's enqueue interface (the CPU test of bench.py's N > 1 control flow)
        if detector_cls is None:
            from .detector import TagDetector
        else:
            TagDetector = detector_cls
        dev = torch.device(device)
        index = dev.index if dev.index is not None else (torch.cuda.current_device() if dev.type == "cuda" else 0)
        self.device = dev
        # (3 or 4 in flight is where the gain levels off; each one holds a full workspace)
        self.depth = min(max(1, int(depth)), 6)
        self.dets = [TagDetector(tag_family, params, device=index) for _ in range(self.depth)]
        # depth 1 stays on the caller's stream (no cross-stream events at all)
        self.streams = [torch.cuda.Stream(dev) for _ in range(self.depth)] if self.depth > 1 else [None]
        self.gather = GatherPipeline(n_frames, dev, dst=dst, group=group, depth=max(2, self.depth),
                                     always_depth=self.depth > 1, slab_records=slab_records, force_collective=force_collective,
                                     every=gather_every,
                                     # (several detectors in flight write their slabs on different streams: a gather of several
                                     # slabs would have to wait for all of them -- one slab per gather there)
                                     steps_per_gather=steps_per_gather if self.depth == 1 else 1)
        self.i = -1
        self.last_table = None

    def submit(self, frames):
        """Enqueue one batch; returns the index of the detector that took it."""
        self.i = (self.i + 1) % self.depth
        det, st = self.dets[self.i], self.streams[self.i]
        if st is None:
            out, table = self.gather.acquire()
            det.saddles_batch_enqueue_to(frames, out, table)
            self.gather.submit()
        else:
            st.wait_stream(torch.cuda.current_stream(self.device))  # the frames' producer
            with torch.cuda.stream(st):
                out, table = self.gather.acquire()
                det.saddles_batch_enqueue_to(frames, out, table)
                self.gather.submit()
        self.last_table = table
        return self.i

    def finish(self):
        """Wait for everything in flight; results of the last submitted batch."""
        res = self.gather.finish()
        for st in self.streams:
            if st is not None:
                st.synchronize()
        return res

    def close(self):
        for d in self.dets:
            d.close()
        self.dets = []


def unpack_frames(saddles, table):
    """Per-frame saddle arrays (numpy, 5 columns x, y, k, theta, phi) from one rank's buffers;
    a frame with a blocking status bit (1|2|4) yields None."""
    s = saddles.cpu().numpy()
    t = table.cpu().numpy()
    out = []
    for count, offset, status, _ in t:
        out.append(None if (status & 7) else s[offset:offset + count].copy())
    return out
