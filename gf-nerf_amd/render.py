"""Full-image inference: Model.get_outputs_for_camera_ray_bundle (nerfstudio/models/base_model.py:163-186) as
scripts/render.py drives it (:109-120), for the GF-NeRF hot path.

The reference loops over chunks of eval_num_rays_per_chunk = 2048 rays (gfnerf/config.py:109) and runs the whole Python
model per chunk: 1013 forwards per 1920x1080 frame, each with several host synchronisations.  Here the rays of the frame
come from one gfn_generate_rays launch, a tile of `tile_rays` rays runs sampler -> hash -> decoder -> compositing with
the live sample count on the device (no synchronisation), and that per-tile launch sequence is captured ONCE in a
hipGraph (torch.cuda.CUDAGraph) over static tile buffers and replayed for every tile.  With `pipeline` = P > 1 one
replay renders P tiles as a producer / consumer pair of graph branches: the (latency-bound) samplers of all P tiles
run back to back on one branch while the (fabric-bound) fused forward of tile i starts as soon as sampler i is done
on the other, its grid capped to `decoder_cus` CUs so the sampler's workgroups find free ones.  Frames shard over ranks
(frame_shard) with one gather of the finished images at the end -- rays are independent, nothing else is exchanged.
"""
import math

import torch

from . import ops


MAX_TILE_RAYS = (1 << 31) // 1024 - 32  # the sampler indexes n_rays * max_hits (1024) in 32 bits


def choose_tile(height, width, target_rays=524288, patch=(8, 4), max_rays=MAX_TILE_RAYS):
    """-> (tile_width, tile_height): a block of pixels that divides the frame exactly (no ragged border tiles), holds whole
    patches, and is close to target_rays and to square (score: |log(rays / target)| + 0.1 |log(width / height)|).  Border tiles padded to full
    size cost a 1920 x 1080 frame 21 % with 512 x 256 tiles.  512 Ki rays: every tile ends with the tail of its march and a hand-over between
    the two branches, and a tile of 400 Ki - 700 Ki rays spreads that over four times the work of the 128 Ki-ray tile rounds 2 - 4 used (full HD:
    960 x 540, 38.6 against 39.9 ms/Mpix; 1920 x 216 / 1920 x 360 / 640 x 1080: 38.1 / 38.1 / 38.4; the record form of the samples costs
    32 KB per ray and buffer: 17 GB each here).  max_rays: no tile above it (the sampler's 32-bit indexing; the caller's memory cap):
    a frame with few divisors (2008 x 1052: 8 x 263) would otherwise get ONE tile of 2.1 M rays."""
    best = None
    target_rays = min(target_rays, max_rays)
    for tw in range(patch[0], width + 1, patch[0]):
        if width % tw:
            continue
        for th in range(patch[1], height + 1, patch[1]):
            if height % th or tw * th > max_rays:
                continue
            key = abs(math.log(tw * th / target_rays)) + 0.1 * abs(math.log(tw / th))
            if best is None or key < best[0]:
                best = (key, tw, th)
    if best is None:  # no exact division into whole patches: fixed-width tiles with ragged borders
        return 128, max(target_rays // 128 // patch[1], 1) * patch[1]
    return best[1], best[2]


def frame_shard(n_frames, world_size, rank):
    """Frames of a camera path rendered by `rank` (round robin: neighbouring frames cost about the same)."""
    return list(range(rank, n_frames, world_size))


class TileRenderer:
    PATCH_MIN_RAYS = 32768  # below ~1 patch of 32 rays per SIMD the patch kernel starves: small tiles keep the wave-per-ray kernel

    def __init__(self, hp, c2w, intr, tile_rays=None, block=None, fineness=1.0, early_T=None, use_graph=True, tile_width=None,
                 pipeline=None, decoder_cus=224, rec_slots=224, hit_cap=128, colour_x2=None):
        """hp: HotPath; c2w f32 [n_cams,3,4] and intr f32 [n_cams,4] = (fx, fy, cx, cy) device tensors of the path's
        cameras; block: focal block table to add (None = global table only); fineness: ray-march fineness of the
        validation mode (rays_noise = ones * fineness, PersSampler_cuda.cu:381-388).
        tile_rays / tile_width: a tile is a tile_width x (tile_rays / tile_width) BLOCK of pixels (tile_width 0: a raster run);
        None = chosen per image size by choose_tile (~512 Ki rays, dividing the frame).  pipeline: tiles per graph replay
        (their samplers run on a second graph branch, through a ring of two sample buffers); None = the largest divisor of the tile
        count <= 16 (a full-HD frame: its 4 tiles in one replay).  rec_slots: record slots per ray, on average, of a ring buffer of the tile
        pipeline (the paged record form, ops.sample_rays_record: 7 KB per ray at 224 instead of the 32 KB of 1024 slots for every ray); a
        frame whose tiles need more is rendered again with a pool half as large again (checked once per frame); 0 or >= 1024: the dense form.
        hit_cap: leaves per ray of the pipeline's hit lists (12 bytes each; the reference's cap, cfg.max_oct_per_ray = 1024, is 12 KB per ray,
        6.4 GB for a 518 k-ray tile, where a ray crosses a few dozen leaves): a frame in which a ray reaches the cap is rendered again with four
        times the cap, up to the reference's -- whose truncation is then the semantics.
        colour_x2: the inference kernel's colour head layer 2 with a one-f16 activation operand (gfn_launch_opts.colour_x2: faster, rgb moves by that layer's
        f16 rounding, density / accumulation / depth do not move at all); None = the engine's HotPathConfig.render_colour_x2."""
        self.colour_x2 = bool(hp.cfg.render_colour_x2 if colour_x2 is None else colour_x2)
        self.hit_cap = max(int(hit_cap), 1)
        self.rec_slots = int(rec_slots) if 0 < int(rec_slots) < 1024 else 0
        self.pool_regrown = 0  # how often a frame had to be rendered again with a larger record pool
        self.hp = hp
        self.c2w, self.intr = (None, None) if c2w is None else (c2w.contiguous(), intr.contiguous())  # None: render_rays only
        self.block, self.early_T, self.fineness = block, early_T, float(fineness)
        self._want = (tile_rays, tile_width, pipeline)
        self.decoder_cus = int(decoder_cus)
        self.use_graph = use_graph
        self.T = None  # buffers are made for the first image (_setup)
        self.graph, self.out = None, None
        if tile_rays is not None:
            self._setup(None, None)

    def _setup(self, height, width):
        """Tile shape, tiles per replay and the static input buffers for images of this size."""
        tile_rays, tile_width, pipeline = self._want
        if tile_rays is None:
            # what a ray costs per ring buffer in the form that WILL be used (paged: rec_slots record slots of 32 bytes + a page table; dense from
            # rec_slots = 0 on: 1024 slots) + the traversal's hit lists of hit_cap leaves: a tile may take a quarter of what is free now (two ring
            # buffers, and the eager warm-ups' cached blocks beside the graph's private pool).  Derived again after every regrow (render_rays)
            free = torch.cuda.mem_get_info(self.hp.dev)[0] if self.hp.dev.type == "cuda" else 1 << 40
            per_ray = (self.rec_slots or 1024) * 32 + 128 + 12 * self.hit_cap
            cap = max(int(free // 4 // per_ray) // 32 * 32, 8192)
            tw, th = choose_tile(height, width, max_rays=min(cap, MAX_TILE_RAYS))
            T = tw * th
        else:
            T = int(tile_rays)
            tw = 128 if tile_width is None else int(tile_width)
            tw = tw if tw and T % tw == 0 else 0
        if pipeline is None:
            if height is None:
                P = 1
            else:
                n_tiles = (-(-height // (T // tw)) * -(-width // tw)) if tw else -(-height * width // T)
                P = max(p for p in range(1, 17) if n_tiles % p == 0)
        else:
            P = max(int(pipeline), 1)
        dev = self.hp.dev
        self.T, self.tw = T, tw
        # neighbouring rays cross the same table cells: gfn_fused_render walks patches of them in lockstep (tile_width passed
        # on); the wave-per-ray fused forward keeps each eighth of a small tile (a strip of rows) on one XCD's L2
        self.patch_tw = tw if T >= self.PATCH_MIN_RAYS else None
        # the samples' form follows the kernel that will read them (ONE predicate, HotPath.uses_patch_kernel: render_patch_width = 0,
        # an exact-f32 decoder or a tile that does not hold whole patches fall back to the wave-per-ray kernel on PACKED samples)
        self.record = self.patch_tw is not None and self.hp.uses_patch_kernel(T, self.patch_tw, self.block)
        if not self.record:
            self.patch_tw = None
        self.noise = torch.full((ops.MAX_SAMPLE_PER_RAY + T + 10,), self.fineness, dtype=torch.float32, device=dev)
        self.P = P if self.use_graph else 1
        self.o = torch.zeros((self.P, T, 3), dtype=torch.float32, device=dev)
        self.d = torch.zeros((self.P, T, 3), dtype=torch.float32, device=dev)
        self.d[..., 2] = -1.0
        self.cam = torch.zeros(T, dtype=torch.int32, device=dev)
        self.graph, self.out = None, None
        self._side = torch.cuda.Stream(device=dev) if self.P > 1 else None
        self._size = (height, width)
        # per frame, read back once at its end: the tiles' overflow words or-ed (a ring buffer's own word is zeroed by the next tile into it)
        self._stat = torch.zeros(1, dtype=torch.int32, device=dev)

    def _tile(self):
        """-> [(rgb, accumulation, depth)] of the P tiles in the input buffers."""
        hp = self.hp
        if self.P == 1:
            s = None
            if self.record:  # the same paged records and hit-list cap as the pipeline's tiles (the memory budget of _setup counts on it)
                s = hp.sample(self.o[0], self.d[0], self.noise, record=True, pool_slots_per_ray=self.rec_slots, max_hits=self.hit_cap)
            out = hp.render(self.o[0], self.d[0], self.cam, self.noise, block=self.block, early_T=self.early_T, samples=s, tile_width=self.patch_tw,
                            opts=ops.launch_opts(colour_x2=self.colour_x2), cam_uniform=True)  # (self.cam holds ONE camera: the image's)
            if s is not None:
                torch.bitwise_or(self._stat, s["overflow"], out=self._stat)
                self._keep = [s]
            return [(out["rgb"], out["accumulation"], out["depth"], out["oct_depth"])]
        main, side = torch.cuda.current_stream(), self._side
        side.wait_stream(main)
        # producer branch: the P samplers back to back (they share one workspace), into a ring of SLOTS sample buffers -- tile i
        # is sampled into the buffer of tile i - SLOTS once that tile's fused forward has read it, so a whole frame can be one
        # replay without one 4.5 GB record buffer per tile; consumer branch: the fused forward of tile i beside the samplers
        # of the tiles after it
        done, used, samples, outs = [], [], [], []
        for i in range(self.P):
            with torch.cuda.stream(side):
                if i >= self.SLOTS:
                    side.wait_event(used[i - self.SLOTS])
                # (tiles for the patch kernel: traversal + counting march only -- the kernel reads the march's records, one copy of the samples)
                samples.append(hp.sample(self.o[i], self.d[i], self.noise, out=samples[i - self.SLOTS] if i >= self.SLOTS else None,
                                         record=self.record, pool_slots_per_ray=self.rec_slots if self.record else 0,
                                         max_hits=self.hit_cap if self.record else None))
                ev = torch.cuda.Event()
                ev.record(side)
                done.append(ev)
            if i >= self.SLOTS - 1:  # (the consumer trails the producer by SLOTS - 1 tiles in enqueue order: a wait needs its event recorded)
                self._consume(i - (self.SLOTS - 1), main, done, used, samples, outs)
        for i in range(max(self.P - (self.SLOTS - 1), 0), self.P):
            self._consume(i, main, done, used, samples, outs)
        main.wait_stream(side)
        self._keep = samples  # sampler outputs were allocated on the side branch: hold them as long as the graph
        return outs

    def footprint_bytes(self):
        """Device memory of this renderer's static buffers: tile inputs, the ring of sample buffers (the record form's march_rec and hit lists
        dominate) and the tiles' outputs -- what the captured graph keeps alive.  None before the first frame (nothing is allocated yet)."""
        if self.T is None or (self.use_graph and self.graph is None):
            return None
        seen, total = set(), 0

        def add(x):
            nonlocal total
            if isinstance(x, torch.Tensor):
                key = (x.untyped_storage().data_ptr(), x.untyped_storage().nbytes())
                if key not in seen:
                    seen.add(key)
                    total += x.untyped_storage().nbytes()
            elif isinstance(x, dict):
                for v in x.values():
                    add(v)
            elif isinstance(x, (list, tuple)):
                for v in x:
                    add(v)

        add([self.o, self.d, self.cam, self.noise, getattr(self, "_keep", None), self.out, self.hp._record_ws, self.hp._sampler_ws])
        return total

    SLOTS = 2  # sample buffers of the tile pipeline: the one the inference kernel reads and the one the next tile's sampler fills

    def _consume(self, i, main, done, used, samples, outs):
        hp = self.hp
        main.wait_event(done[i])
        # (the inference kernel's grid is capped -- baked into the captured launch -- so that the next tile's sampler finds free CUs)
        out = hp.render(self.o[i], self.d[i], self.cam, self.noise, block=self.block, early_T=self.early_T, samples=samples[i], tile_width=self.patch_tw,
                        opts=ops.launch_opts(decoder_cus=self.decoder_cus, colour_x2=self.colour_x2), cam_uniform=True)
        outs.append((out["rgb"], out["accumulation"], out["depth"], out["oct_depth"]))
        if self.record:
            torch.bitwise_or(self._stat, samples[i]["overflow"], out=self._stat)
        ev = torch.cuda.Event()
        ev.record(main)
        used.append(ev)

    def _capture(self):
        """Warm up on a side stream (workspaces, kernel attributes, weight images), then capture one tile."""
        s = torch.cuda.Stream(device=self.hp.dev)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._tile()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._tile()

    @torch.no_grad()
    def render_camera(self, cam, height, width, embed_idx=None):
        """One image of camera `cam` -> dict(rgb [H,W,3], accumulation [H,W], depth [H,W]) on the device.
        embed_idx: appearance-embedding row (render.py --embedding-indices); default = the camera's own index."""
        n = height * width
        rays = ops.generate_rays(self.c2w, self.intr, cam0=int(cam), first_pixel=0, width=int(width), n=n, want_extras=False)
        out = self.render_rays(rays["origins"], rays["directions"], int(cam if embed_idx is None else embed_idx), height, width)
        return dict(rgb=out["rgb"], accumulation=out["accumulation"].view(height, width), depth=out["depth"].view(height, width))

    @torch.no_grad()
    def render_rays(self, O, D, embed_idx, height, width):
        """The rays of one image, row major: O, D f32 [H*W,3] on the device; embed_idx: the appearance-embedding row of the
        whole image -> dict(rgb [H,W,3], accumulation / depth / oct_depth [H,W,1])."""
        for _ in range(8):
            out = self._render_rays(O, D, embed_idx, height, width)
            if not self.record:
                return out
            ovf = int(self._stat.item())  # (one synchronisation per frame: ~0.1 % of a full-HD frame)
            hits_short = bool(ovf & ops.HIT_CAP_REACHED) and self.hit_cap < self.hp.cfg.max_oct_per_ray
            if not (ovf & ops.REC_POOL_OVERFLOW) and not hits_short:
                return out
            # a tile's rays needed more record pages than the pool holds: a pool half as large again (the dense form from 1024 slots per ray on);
            # or a ray crossed more leaves than the hit lists hold: four times the cap.  Captured again, and this frame once more
            if ovf & ops.REC_POOL_OVERFLOW:
                self.rec_slots = (int(self.rec_slots * 1.5) + 31) // 32 * 32
                if self.rec_slots >= 1024:
                    self.rec_slots = 0  # (the dense form: no pages)
            if hits_short:
                self.hit_cap = min(self.hit_cap * 4, self.hp.cfg.max_oct_per_ray)
                self.hp._record_ws = None  # (hit lists of the old stride)
            self.pool_regrown += 1
            self.graph, self.out, self._keep = None, None, None
            torch.cuda.empty_cache()
            if self._want[0] is None:
                self.T = None  # the tile size follows the new per-ray cost and what is free NOW (_setup): smaller tiles instead of running out of memory
        raise RuntimeError("TileRenderer: the record pool kept overflowing")

    def _render_rays(self, O, D, embed_idx, height, width):
        if self.T is None or (self._want[0] is None and self._size != (height, width)):
            self._setup(height, width)
        hp, T = self.hp, self.T
        n = height * width
        O, D = O.reshape(n, 3), D.reshape(n, 3)
        self.cam.fill_(int(embed_idx))
        rgb = torch.empty((n, 3), dtype=torch.float32, device=hp.dev)
        acc = torch.empty((n, 1), dtype=torch.float32, device=hp.dev)
        depth = torch.empty((n, 1), dtype=torch.float32, device=hp.dev)
        octd = torch.empty((n, 1), dtype=torch.float32, device=hp.dev)
        if self.use_graph and self.graph is None:
            self.o.copy_((O[:T] if n >= T else torch.cat([O, O[-1:].expand(T - n, 3)]))[None].expand_as(self.o))
            self.d.copy_((D[:T] if n >= T else torch.cat([D, D[-1:].expand(T - n, 3)]))[None].expand_as(self.d))
            self._capture()
        self._stat.zero_()  # (behind the capture's eager warm-up tiles: the statistics are this frame's)
        if self.tw:
            tw, th = self.tw, T // self.tw
            O2, D2 = O.view(height, width, 3), D.view(height, width, 3)
            rgb2, acc2, dep2, oct2 = rgb.view(height, width, 3), acc.view(height, width, 1), depth.view(height, width, 1), octd.view(height, width, 1)
            tiles = [(y, min(y + th, height), x, min(x + tw, width)) for y in range(0, height, th) for x in range(0, width, tw)]
        else:
            tiles = [(a, min(a + T, n), 0, 0) for a in range(0, n, T)]
        P = self.P
        for g0 in range(0, len(tiles), P):
            group = tiles[g0:g0 + P]
            group = group + [group[-1]] * (P - len(group))  # a short last group renders its last tile again
            sizes = []
            for i, (a, b, x0, x1) in enumerate(group):
                o, d = self.o[i], self.d[i]
                if self.tw:
                    h, w = b - a, x1 - x0
                    m = h * w
                    if m == T:
                        o.view(h, w, 3).copy_(O2[a:b, x0:x1]); d.view(h, w, 3).copy_(D2[a:b, x0:x1])
                    else:  # ragged border tile
                        o[:m].copy_(O2[a:b, x0:x1].reshape(m, 3)); d[:m].copy_(D2[a:b, x0:x1].reshape(m, 3))
                else:
                    m = b - a
                    o[:m].copy_(O[a:b]); d[:m].copy_(D[a:b])
                if m < T:  # ragged tile: repeat its last ray, the extra outputs are dropped
                    o[m:].copy_(o[m - 1:m].expand(T - m, 3)); d[m:].copy_(d[m - 1:m].expand(T - m, 3))
                sizes.append(m)
            if self.use_graph:
                self.graph.replay()
                outs = self.out
            else:
                outs = self._tile()
            for i, (a, b, x0, x1) in enumerate(tiles[g0:g0 + P]):
                t_rgb, t_acc, t_depth, t_oct = outs[i]
                m = sizes[i]
                if self.tw:
                    h, w = b - a, x1 - x0
                    rgb2[a:b, x0:x1] = t_rgb[:m].view(h, w, 3); acc2[a:b, x0:x1] = t_acc[:m].view(h, w, 1); dep2[a:b, x0:x1] = t_depth[:m].view(h, w, 1)
                    oct2[a:b, x0:x1] = t_oct[:m].view(h, w, 1)
                else:
                    rgb[a:b].copy_(t_rgb[:m]); acc[a:b].copy_(t_acc[:m]); depth[a:b].copy_(t_depth[:m]); octd[a:b].copy_(t_oct[:m])
        return dict(rgb=rgb.view(height, width, 3), accumulation=acc.view(height, width, 1), depth=depth.view(height, width, 1),
                    oct_depth=octd.view(height, width, 1))

    def render_path(self, cams, height, width, world_size=1, rank=0, owners=None):
        """Camera path sharded over ranks by frame -> {frame index: outputs} of this rank's frames.  owners[f] (see
        gfnerf_amd.dist.frame_owner) routes frame f to the rank holding its block's table; default: round robin."""
        mine = frame_shard(len(cams), world_size, rank) if owners is None else [f for f in range(len(cams)) if owners[f] == rank]
        return {f: self.render_camera(cams[f], height, width) for f in mine}


def gather_frames(local, n_frames, height, width, dst=0, owners=None):
    """Collect every rank's rgb frames on `dst` -> [n_frames,H,W,3] (None elsewhere).  One collective per path.
    owners[f]: the rank that rendered frame f (gfnerf_amd.dist.frame_owner: the owner of the frame's block); default: round
    robin (frame_shard)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return torch.stack([local[f]["rgb"] for f in range(n_frames)])
    w, r = dist.get_world_size(), dist.get_rank()
    if owners is None:
        owners = [f % w for f in range(n_frames)]
    frames_of = [[f for f in range(n_frames) if owners[f] == rr] for rr in range(w)]
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if local:
        dev = next(iter(local.values()))["rgb"].device
    per = max(max(len(fr) for fr in frames_of), 1)  # gather needs equal shapes: pad to the busiest rank
    mine = torch.zeros((per, height, width, 3), dtype=torch.float32, device=dev)
    for k, f in enumerate(frames_of[r]):
        mine[k] = local[f]["rgb"]
    bufs = [torch.empty_like(mine) for _ in range(w)] if r == dst else None
    dist.gather(mine, bufs, dst=dst)
    if r != dst:
        return None
    out = torch.empty((n_frames, height, width, 3), dtype=torch.float32, device=dev)
    for rr in range(w):
        for k, f in enumerate(frames_of[rr]):
            out[f] = bufs[rr][k]
    return out
