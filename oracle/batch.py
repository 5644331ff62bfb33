"""CPU restatement (numpy) of batch assembly -- TEST INFRASTRUCTURE ONLY, never imported by the product.

Follows nerfstudio/data/pixel_samplers.py (PixelSampler :47-117, ErrorPixelSampler :636-723),
nerfstudio/data/utils/dataloaders.py:140-142 and gfnerf/gf_pipeline.py:180-185.

Pinned by `tests/golden/reference_vectors.npz` (`pix_*`, `errpix_*` arrays, captured from the reference's own classes by
tests/golden/make_golden.py): the uniform index arithmetic, the collate gathers and the error-map write are
deterministic given the random numbers and are held exactly; the weighted sampler draws from different random streams
(torch.multinomial's exponential race there, inverse CDF + duplicate rejection here -- both are sequential sampling
without replacement with probability proportional to the weights), so it is pinned in distribution: inclusion
frequencies of the reference over many seeded calls.
"""
import numpy as np

ERRMAP_BLOCK = 4096
PIXEL_CHUNK = 1 << 24
PIXEL_ATTEMPTS = 64
_M64 = (1 << 64) - 1


def pixel_uniform(u, n_images, H, W):
    """pixel_samplers.py:63-67: floor(rand * tensor([n, H, W])).long(); f32 product; clamped where the reference would
    index out of range.  -> lin i64 [n]."""
    u = np.asarray(u, np.float32)
    ext = np.array([n_images, H, W], np.float32)
    idx = np.floor(u * ext).astype(np.int64)
    idx = np.minimum(np.maximum(idx, 0), np.array([n_images - 1, H - 1, W - 1]))
    return (idx[:, 0] * H + idx[:, 1]) * W + idx[:, 2]


def pixel_collate(lin, images, image_idx=None, rel_cam_idx=None):
    """pixel_samplers.py:73-103 (+ ray_generators.py:47-52 coords).  images f32 or u8 [n,H,W,3]."""
    n, H, W = images.shape[:3]
    lin = np.asarray(lin, np.int64)
    x = lin % W
    y = (lin // W) % H
    c = lin // (W * H)
    px = images[c, y, x]
    if images.dtype == np.uint8:
        px = px.astype(np.float32) / np.float32(255.0)  # base_dataset.get_image
    cam = c if image_idx is None else np.asarray(image_idx)[c]
    rel = c if rel_cam_idx is None else np.asarray(rel_cam_idx)[c]
    return dict(indices=np.stack([cam, y, x], 1).astype(np.int32), slot=c.astype(np.int32), rel_camera_indices=rel.astype(np.int32),
                image=px.astype(np.float32), coords=np.stack([y + 0.5, x + 0.5], 1).astype(np.float32))


def errmap_block_sums(w):
    w = np.asarray(w, np.float32).reshape(-1)
    nb = (w.size + ERRMAP_BLOCK - 1) // ERRMAP_BLOCK
    pad = np.zeros(nb * ERRMAP_BLOCK, np.float64)
    pad[: w.size] = w
    return pad.reshape(nb, ERRMAP_BLOCK).sum(1)


def errmap_update(error_map, indices, pred, gt):
    """gf_pipeline.py:181-184 + dataloaders.py:142, in place on error_map f32 [n,H,W(,1)]; rows out of range are skipped
    (the reference raises)."""
    pred, gt = np.asarray(pred, np.float32), np.asarray(gt, np.float32)
    e = np.abs(gt - pred)
    e = (e[:, 0] + e[:, 1]) + e[:, 2]
    n, H, W = error_map.shape[:3]
    flat = error_map.reshape(n, H, W)
    for (c, y, x), v in zip(np.asarray(indices), e):
        if 0 <= c < n and 0 <= y < H and 0 <= x < W:
            flat[c, y, x] = v
    return e


def draw_range(d, n_weighted, n):
    """Which pixels draw d samples from: weighted_choice_multinomial's chunk quotas and left-over rule (:636-674)."""
    if d >= n_weighted:
        return 0, n, False
    nc = n // PIXEL_CHUNK
    if nc == 0:
        return 0, n, True
    per = n_weighted // nc
    if d < per * nc:
        c = d // per
        return c * PIXEL_CHUNK, (c + 1) * PIXEL_CHUNK, True
    if n % nc != 0:
        return nc * PIXEL_CHUNK, n, True
    return 0, n, False


def _candidate(w, prefix, bits, lo, hi, weighted):
    r64 = int(bits) & 0xFFFFFFFFFFFFFFFF
    b0, b1 = lo // ERRMAP_BLOCK, (hi + ERRMAP_BLOCK - 1) // ERRMAP_BLOCK
    p0 = prefix[b0]
    total = prefix[b1] - p0
    if not weighted or not total > 0.0:
        return lo + ((r64 * (hi - lo)) >> 64), bool(weighted)
    u = float(r64 >> 11) * 2.0 ** -53
    target = p0 + u * total
    b = int(np.searchsorted(prefix[b0:b1], target, side="right")) - 1 + b0
    while b > b0 and not prefix[b + 1] > prefix[b]:
        b -= 1
    while b + 1 < b1 and not prefix[b + 1] > prefix[b]:
        b += 1
    r = target - prefix[b]
    base = b * ERRMAP_BLOCK
    seg = np.zeros(ERRMAP_BLOCK, np.float64)
    a, z = max(base, lo), min(base + ERRMAP_BLOCK, hi, w.size)
    seg[a - base: z - base] = w[a:z]
    # the device's association: lane l adds its 64 consecutive elements in order, the lane totals are added in lane
    # order, and each lane then runs (excl + v0) + v1 + ... (np.cumsum is sequential)
    lane = seg.reshape(64, 64)
    sums = np.cumsum(lane, axis=1)[:, -1]
    excl = np.concatenate([[0.0], np.cumsum(sums)[:-1]])
    run = np.cumsum(np.concatenate([excl[:, None], lane], axis=1), axis=1)[:, 1:].reshape(-1)
    hit = np.nonzero((seg > 0) & (run > r))[0]
    if hit.size:
        pick = int(hit[0])
    else:
        pos = np.nonzero(seg > 0)[0]
        pick = int(pos[-1]) if pos.size else 0
    return base + pick, False


def attempt_bits(seed, a):
    """splitmix64(seed + (a + 1) * golden): the 64 bits attempt a of a draw uses."""
    z = ((int(seed) & _M64) + (a + 1) * 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def errmap_prefix(sums):
    """The device's prefix of the block sums: 1024 sequential segments whose totals are scanned sequentially."""
    nb = sums.size
    per = (nb + 1023) // 1024
    prefix = np.zeros(nb + 1)
    run0 = 0.0
    for t in range(1024):
        a, b = min(t * per, nb), min(t * per + per, nb)
        if b > a:
            prefix[a + 1: b + 1] = np.cumsum(np.concatenate([[run0], sums[a:b]]))[1:]
            run0 += np.cumsum(sums[a:b])[-1]
    return prefix


def errmap_sample(error_map, bits, n_weighted, block_sums=None):
    """ErrorPixelSampler.sample_method (:676-723) as the device computes it.  bits i64 [n], one word per draw.
    -> (lin i64 [n], status [2]).  Without replacement = a draw whose candidate is already held (by an earlier attempt, or
    by a lower draw of the same attempt) draws again; the weighted and the uniform group are independent samples."""
    w = np.asarray(error_map, np.float32).reshape(-1)
    bits = np.asarray(bits, np.int64).reshape(-1)
    n = bits.size
    sums = errmap_block_sums(w) if block_sums is None else np.asarray(block_sums, np.float64)
    prefix = errmap_prefix(sums)
    status = [0, 0]
    lin = np.full(n, -1, np.int64)
    cand = np.zeros(n, np.int64)
    for d in range(n):
        lo, hi, weighted = draw_range(d, n_weighted, w.size)
        cand[d], zero = _candidate(w, prefix, attempt_bits(bits[d], 0), lo, hi, weighted)
        status[0] += int(zero)
    for g0, g1 in ((0, n_weighted), (n_weighted, n)):
        taken = set()
        for a in range(PIXEL_ATTEMPTS):
            new = {}
            for d in range(g0, g1):  # lowest draw of this attempt wins; earlier attempts always win
                if lin[d] < 0 and cand[d] not in taken and cand[d] not in new:
                    new[cand[d]] = d
            for key, d in new.items():
                lin[d] = key
            taken.update(new.keys())
            pending = [d for d in range(g0, g1) if lin[d] < 0]
            if not pending:
                break
            if a + 1 < PIXEL_ATTEMPTS:
                for d in pending:
                    lo, hi, weighted = draw_range(d, n_weighted, w.size)
                    cand[d], _ = _candidate(w, prefix, attempt_bits(bits[d], a + 1), lo, hi, weighted)
        for d in range(g0, g1):
            if lin[d] < 0:
                lin[d] = cand[d]
                status[1] += 1
    return lin, np.array(status, np.int32)
