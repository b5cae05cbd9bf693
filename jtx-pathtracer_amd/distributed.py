"""Multi-GPU pixel-tile sharding: one process per GPU, torch.distributed (RCCL on ROCm, gloo on CPU).

Every pixel-sample is an independent path whose RNG depends only on (row, col, sample)
(camera.cpp:101), and the reference already partitions the frame into 32x32 tiles with no inter-tile
communication (camera.cpp:55-64).  Rank r owns the row-major 32x32 tiles k with k % world == r
(interleaved, so sky-heavy and geometry-heavy regions spread over all GPUs) and accumulates ALL
strata of its own pixels in the reference's sample order; the per-pixel float sums are therefore
bit-identical to the 1-GPU result.  The scene is replicated.  The only collective is one exchange per
frame that brings the disjoint shards to rank 0, in one of two forms:
  * FrameGather (default): every rank packs its OWN pixels (12 B accumulation + 3 B RGB8 each) into a
    compact slab and one gather delivers the slabs to rank 0 -- xGMI is point-to-point, so the 7 slabs of an
    8-GPU node travel on 7 different links at once and each link carries 1/8 of the frame;
  * reduce_frame: one sum-reduce of the full-size buffers in which non-owned pixels are exactly zero
    (a ring: every link carries ~7/8 of the frame).
"""
import ctypes as C

import numpy as np

from . import _capi as capi

TILE = 32
FRAME_SLOTS = 3            # jtx_mi.h: JTX_MI_FRAME_SLOTS


def tile_owner_mask(width, height, rank, world):
    """Boolean (H, W) mask of the pixels whose 32x32 tile belongs to `rank`."""
    tiles_x = (width + TILE - 1) // TILE
    rows = np.arange(height)[:, None] // TILE
    cols = np.arange(width)[None, :] // TILE
    return ((rows * tiles_x + cols) % max(1, world)) == rank


def owned_tiles(width, height, rank, world):
    tiles = ((width + TILE - 1) // TILE) * ((height + TILE - 1) // TILE)
    return len(range(rank, tiles, max(1, world)))


def render_shard(scene, cam_desc, rank, world, acc, img=None, stream=None, count_rays=False,
                 sample_begin=0, sample_end=0, integrator=0, profile_kernels=False, frame_slot=0, sequence_end=False, max_record_mb=0):
    """Launch this rank's share of the frame into device tensors `acc` (H*W*3 f32) / `img` (H*W*3 u8).

    Asynchronous on `stream` (an int hipStream_t, e.g. torch.cuda.current_stream().cuda_stream).  `frame_slot` 0 .. FRAME_SLOTS - 1: that
    many frames of one scene may be in flight, one per slot, each on a stream and into a pair of film buffers of its own (jtx_mi.h:
    jtx_mi_render_opts).  Launches that use the scene's singletons instead of a slot's working memory -- count_rays, the wavefront
    integrator, the alternate Li, JTX_DYNAMIC_PATHS=0 -- are ordered against EVERY slot by the library.  `max_record_mb`: the cap on one
    launch's radiance records (0: 8 GiB; a frame above it goes in several launches of consecutive strata).
    """
    lib = capi.load()
    o = capi.RenderOpts()
    o.tile_rank, o.tile_world = rank, world
    o.count_rays = 1 if count_rays else 0
    o.sample_begin, o.sample_end = sample_begin, sample_end
    o.integrator = integrator
    o.reserved = 1 if profile_kernels else 0
    o.frame_slot = frame_slot
    o.sequence_end = 1 if sequence_end else 0
    o.max_record_mb = max_record_mb
    capi.check(lib.jtx_mi_render_device(scene.handle, C.byref(cam_desc), C.byref(o),
                                        C.c_void_p(acc.data_ptr()),
                                        C.c_void_p(img.data_ptr()) if img is not None else None,
                                        C.c_void_p(stream) if stream else None))


def _gloo_with_device_tensors(t, group=None):
    """gloo (CPU rehearsals of the N > 1 path on a one-GPU box) has no reduce / gather for device tensors"""
    import torch.distributed as dist
    return t.is_cuda and dist.get_backend(group) == "gloo"


def reduce_frame(acc, img=None, dst=0, group=None):
    """The per-frame collective: sum the disjoint shards onto rank `dst` (RCCL ring over xGMI)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in (acc, img):
        if t is None:
            continue
        if _gloo_with_device_tensors(t, group):
            h = t.cpu()
            dist.reduce(h, dst=dst, op=dist.ReduceOp.SUM, group=group)
            if dist.get_rank(group) == dst:
                t.copy_(h)
        else:
            dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)


class FrameGather:
    """Per-frame exchange by compact slabs: pack own pixels -> gather to `dst` -> scatter into the frame.

    Built once per (width, height, world); `collect(acc, img)` is called after render_shard on the same
    stream.  acc: H*W*3 float32, img: H*W*3 uint8 device (or CPU/gloo) tensors; on `dst` they hold the
    whole frame afterwards, bit for bit what one rank renders alone.
    """

    def __init__(self, width, height, rank, world, device, dst=0, group=None):
        import torch
        self.rank, self.world, self.dst, self.group = rank, world, dst, group
        self.npix = width * height
        idx = [np.flatnonzero(tile_owner_mask(width, height, r, world).reshape(-1)) for r in range(world)]
        self.count = [len(i) for i in idx]
        self.nmax = max(1, max(self.count))
        pad = lambda a: np.concatenate([a, np.full(self.nmax - len(a), a[-1] if len(a) else 0, np.int64)])
        self.own = torch.from_numpy(pad(idx[rank]).astype(np.int64)).to(device)
        self.slab = torch.zeros((self.nmax, 15), dtype=torch.uint8, device=device)
        if rank == dst:
            # padded entries repeat the rank's last own pixel (same bytes written twice); a rank without
            # pixels is dropped from the scatter altogether
            keep = [r for r in range(world) if self.count[r] > 0]
            self.keep = torch.tensor(keep, dtype=torch.int64, device=device)
            self.all_idx = torch.from_numpy(np.concatenate([pad(idx[r]) for r in keep]).astype(np.int64)).to(device) \
                if keep else None
            self.recv = torch.zeros((world, self.nmax, 15), dtype=torch.uint8, device=device)

    def pack(self, acc, img):
        """own pixels of (acc, img) -> self.slab [nmax, 15] uint8"""
        import torch
        a8 = acc.view(torch.uint8).view(self.npix, 12)
        i8 = img.view(self.npix, 3)
        self.slab[:, :12].copy_(a8.index_select(0, self.own))
        self.slab[:, 12:].copy_(i8.index_select(0, self.own))
        return self.slab

    def scatter(self, acc, img):
        """(dst only) self.recv [world, nmax, 15] -> the frame buffers (the shard buffers themselves, or separate ones)"""
        import torch
        if self.all_idx is None:
            return
        a8 = acc.view(torch.uint8).view(self.npix, 12)
        i8 = img.view(self.npix, 3)
        got = self.recv.index_select(0, self.keep).view(-1, 15)
        a8.index_copy_(0, self.all_idx, got[:, :12].contiguous())
        i8.index_copy_(0, self.all_idx, got[:, 12:].contiguous())

    def collect(self, acc, img, out_acc=None, out_img=None):
        """pack -> gather -> scatter.  On `dst` the frame lands in (out_acc, out_img) if given, else in place in
        (acc, img).  Everything is enqueued on the CURRENT torch stream, so a caller may run it on a side stream
        while the next frame renders into another pair of shard buffers."""
        import torch.distributed as dist
        if self.world == 1 or not dist.is_initialized():
            return
        self.pack(acc, img)
        if out_acc is not None:
            acc, img = out_acc, out_img
        if _gloo_with_device_tensors(self.slab, self.group):             # rehearsal only: stage through the host
            slab = self.slab.cpu()
            if self.rank == self.dst:
                recv = [slab.new_empty(slab.shape) for _ in range(self.world)]
                dist.gather(slab, recv, dst=self.dst, group=self.group)
                for r in range(self.world):
                    self.recv[r].copy_(recv[r])
                self.scatter(acc, img)
            else:
                dist.gather(slab, None, dst=self.dst, group=self.group)
            return
        if self.rank == self.dst:
            dist.gather(self.slab, list(self.recv.unbind(0)), dst=self.dst, group=self.group)
            self.scatter(acc, img)
        else:
            dist.gather(self.slab, None, dst=self.dst, group=self.group)


class ShardPipeline:
    """Frame loop of one rank with SEVERAL FRAMES IN FLIGHT: frame i + 1 is launched on another render stream, into another pair of
    shard buffers and another of the scene's frame slots (jtx_mi_render_opts.frame_slot), while frame i's last chunks still run --
    every persistent wave ends with ever fewer live lanes: ~0.4 ms, 10 % of a 1/8 shard --, and frame i's resolve pass, which finds
    no free wave slot while frame i + 1 fills the chip, runs beside the first chunks of frame i + 2 (hence three slots); with a
    `gatherer` (N > 1) frame i is then packed and gathered on a side stream, and on `dst` frames are assembled in buffers of their
    own (`frame_acc`, `frame_img`).  The workers of StaticCamera::render never wait for a frame boundary either (camera.cpp:53-64, 81-123).
    `step()` only enqueues; `torch.cuda.synchronize()` (or the events) completes it.  Without a gatherer (one rank) the finished frame
    of step k is in `accs[k % len(accs)]` / `imgs[k % len(accs)]`.  `timing=True` (or start_timing()) keeps a pair of timing events per frame around
    the exchange for exchange_ms(); off by default -- a long-running loop must not pile up live events nobody reads."""

    def __init__(self, scene, cam_desc, rank, world, device, gatherer=None, integrator=0, timing=False, frames_in_flight=FRAME_SLOTS,
                 deliver_to_host=False):
        import torch
        self.scene, self.cam, self.rank, self.world, self.integrator = scene, cam_desc, rank, world, integrator
        self.gatherer = gatherer
        n = cam_desc.width * cam_desc.height * 3
        nb = max(2, min(FRAME_SLOTS, int(frames_in_flight)))          # buffer sets: one per frame in flight (two also for one stream: the exchange reads one)
        self.accs = [torch.zeros(n, dtype=torch.float32, device=device) for _ in range(nb)]
        self.imgs = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(nb)]
        on_dst = gatherer is not None and rank == gatherer.dst
        self.frame_acc = torch.zeros(n, dtype=torch.float32, device=device) if on_dst else None
        self.frame_img = torch.zeros(n, dtype=torch.uint8, device=device) if on_dst else None
        # frames_in_flight = 1: both buffer pairs on ONE render stream and one frame slot (round 4's loop: only the exchange overlaps)
        self.rstreams = [torch.cuda.Stream(device=device) for _ in range(nb if frames_in_flight > 1 else 1)]
        self.xstream = torch.cuda.Stream(device=device) if gatherer is not None else None
        self.rendered = [torch.cuda.Event() for _ in range(nb)]
        self.exchanged = [torch.cuda.Event() for _ in range(nb)]
        self.n = 0
        self.timing = bool(timing)
        self.timed = []                  # (timing only) per frame: (event before, event after) the exchange on the side stream
        self._ms, self._nms = 0.0, 0
        # deliver_to_host (a frame is done when the last byte of acc / img is on the HOST): every finished frame -- this rank's
        # film (one rank) or the assembled frame (the gather's destination) -- is copied to page-locked host buffers, one pair per buffer
        # set, on a copy stream of its own (one rank) or behind the exchange on its stream; a buffer set is rendered into again only when
        # its copy has left.  Frame k is in host_acc[k % len(host_acc)] / host_img[...] once `copied[...]` has passed.
        self.deliver = bool(deliver_to_host) and (gatherer is None or on_dst)
        if self.deliver:
            self.host_acc = [torch.empty(n, dtype=torch.float32, pin_memory=True) for _ in range(nb)]
            self.host_img = [torch.empty(n, dtype=torch.uint8, pin_memory=True) for _ in range(nb)]
            self.cstream = torch.cuda.Stream(device=device) if gatherer is None else None
            self.copied = [torch.cuda.Event() for _ in range(nb)]

    def prime(self):
        """one untimed frame per frame slot, then a synchronisation: every slot's radiance records (2 GB each for a 1080p x 64 spp frame)
        are allocated and their pages mapped at first touch -- 6 ms per slot that belong to set-up, not to the first frames of a loop"""
        import torch
        for _ in range(len(self.accs)):
            self.step(last=True)
        torch.cuda.synchronize()
        self.n = 0
        self.reset_timing()

    def start_timing(self):
        self.timing = True
        self.reset_timing()

    def reset_timing(self):
        self.timed = []
        self._ms, self._nms = 0.0, 0

    def _fold(self, keep=0):
        """completed event pairs -> running sum (bounded memory: at most `keep` + the pairs still in flight stay alive)"""
        while len(self.timed) > keep and self.timed[0][1].query():
            a, b = self.timed.pop(0)
            self._ms += a.elapsed_time(b); self._nms += 1

    def exchange_ms(self):
        """average device time of one frame's exchange (pack + gather + scatter on the side stream) since reset_timing() /
        start_timing(); call after torch.cuda.synchronize().  None before the first timed frame."""
        self._fold()
        return self._ms / self._nms if self._nms else None

    def wait(self, stream):
        """make `stream` wait for everything enqueued so far (both render streams and the exchange)"""
        import torch
        for st in self.rstreams + ([self.xstream] if self.xstream is not None else []):
            e = torch.cuda.Event(); e.record(st); stream.wait_event(e)

    def step(self, render_stream=None, last=False):
        """enqueue one frame.  `last`: nothing follows it (jtx_mi_render_opts.sequence_end: its launch is cut like a lone one, so that its
        end is short).  (`render_stream`: accepted for round 4's signature; the pipeline renders on streams of its own)"""
        import torch
        b = self.n % len(self.accs)
        self.n += 1
        rs = self.rstreams[b % len(self.rstreams)]
        if self.gatherer is not None:
            rs.wait_event(self.exchanged[b])                 # the exchange two frames back has read this pair
        elif self.deliver:
            rs.wait_event(self.copied[b])                    # the frame that was in this pair has left for the host
        render_shard(self.scene, self.cam, self.rank, self.world, self.accs[b], self.imgs[b],
                     stream=rs.cuda_stream, integrator=self.integrator, frame_slot=b % len(self.rstreams), sequence_end=last)
        self.rendered[b].record(rs)
        if self.gatherer is None:
            if self.deliver:
                self.cstream.wait_event(self.rendered[b])
                with torch.cuda.stream(self.cstream):
                    self.host_acc[b].copy_(self.accs[b], non_blocking=True)
                    self.host_img[b].copy_(self.imgs[b], non_blocking=True)
                    self.copied[b].record(self.cstream)
            return
        with torch.cuda.stream(self.xstream):
            self.xstream.wait_event(self.rendered[b])
            if self.timing:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(self.xstream)
            self.gatherer.collect(self.accs[b], self.imgs[b], self.frame_acc, self.frame_img)
            if self.timing:
                t1.record(self.xstream)
                self.timed.append((t0, t1))
                self._fold(keep=64)
            self.exchanged[b].record(self.xstream)
            if self.deliver:                                 # the assembled frame, behind its exchange (the next exchange scatters into the same buffers)
                self.host_acc[b].copy_(self.frame_acc, non_blocking=True)
                self.host_img[b].copy_(self.frame_img, non_blocking=True)
                self.copied[b].record(self.xstream)
