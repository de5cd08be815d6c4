"""Seeded synthetic inputs in the reference's data formats.

No KITTI/TUM data exists offline, so the harness that stands in for the reference's
dataset loaders (Examples/Stereo/stereo_kitti.cc:81-155, Examples/RGB-D/rgbd_my.cc:86-131,
185-254) generates frames of the same shape:

  * stereo 1241x376 u8 pairs: a rectangle/disc texture over value noise, warped by a
    piecewise-constant integer disparity (6 vertical bands, 4..64 px);
  * RGB-D: the left image as 3-channel u8 + depth u16 = round(DepthMapFactor*bf/disparity);
  * sequence motion: frame t+1 = frame t translated by (3,0) px and scaled 1.01 about the
    principal point (forward motion), timestamps 0.1 s apart;
  * boxes: `id cx cy w h` rows (rgbd_my.cc:246-249); masks: u8 {0,255} ellipses in boxes.

Only numpy; the PCG64 integer/uniform streams used here are stable across numpy versions.
"""
import numpy as np

BASE_SEED = 0x51A3D1C

# Examples/Stereo/KITTI04-12.yaml:8-51 and Examples/RGB-D/KITTI03.yaml (values only)
KITTI_STEREO = dict(width=1241, height=376, fx=707.0912, fy=707.0912, cx=601.8873, cy=183.1104,
                    bf=379.8145, th_depth=40.0, fps=10.0, n_features=2000, scale_factor=1.2,
                    n_levels=8, ini_th_fast=12, min_th_fast=7)
KITTI03_RGBD = dict(width=1241, height=376, fx=721.5377, fy=721.5377, cx=609.5593, cy=172.854,
                    bf=387.5744, th_depth=40.0, fps=10.0, n_features=2000, scale_factor=1.2,
                    n_levels=8, ini_th_fast=20, min_th_fast=7, depth_map_factor=1000.0)
TUM3 = dict(width=640, height=480, fx=535.4, fy=539.2, cx=320.1, cy=247.6, bf=40.0, th_depth=40.0,
            fps=30.0, n_features=1000, scale_factor=1.2, n_levels=8, ini_th_fast=20, min_th_fast=7,
            depth_map_factor=5000.0)


# Examples/RGB-D/TUM1.yaml: the one family of shipped settings with lens distortion (Camera.k1 != 0)
TUM1 = dict(width=640, height=480, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989, k1=0.262383, k2=-0.953104, p1=-0.005358,
            p2=0.002628, k3=1.163314, bf=40.0, th_depth=40.0, fps=30.0, n_features=1000, scale_factor=1.2, n_levels=8, ini_th_fast=20,
            min_th_fast=7, depth_map_factor=5000.0)


def _rng(seq, frame):
    return np.random.Generator(np.random.PCG64(BASE_SEED + 1000 * seq + frame))


def base_texture(width, height, seq=0, margin=96):
    """Large texture canvas the frames of one sequence are cut from."""
    rng = _rng(seq, 999)
    W, H = width + 2 * margin, height + 2 * margin
    img = np.full((H, W), 128, np.int16)
    n_rect = int(4000 * (W * H) / (1433 * 568))
    n_disc = int(600 * (W * H) / (1433 * 568))
    xs = rng.integers(0, W, n_rect); ys = rng.integers(0, H, n_rect)
    ws = rng.integers(4, 60, n_rect); hs = rng.integers(4, 40, n_rect)
    vs = rng.integers(20, 236, n_rect)
    for x, y, w, h, v in zip(xs, ys, ws, hs, vs):
        img[y:y + h, x:x + w] = v
    cx = rng.integers(0, W, n_disc); cy = rng.integers(0, H, n_disc)
    rr = rng.integers(3, 14, n_disc); vs = rng.integers(20, 236, n_disc)
    yy, xx = np.mgrid[-14:15, -14:15]
    for x, y, r, v in zip(cx, cy, rr, vs):
        y0, y1, x0, x1 = max(y - 14, 0), min(y + 15, H), max(x - 14, 0), min(x + 15, W)
        m = (yy[y0 - y + 14:y1 - y + 14, x0 - x + 14:x1 - x + 14] ** 2 +
             xx[y0 - y + 14:y1 - y + 14, x0 - x + 14:x1 - x + 14] ** 2) <= r * r
        img[y0:y1, x0:x1][m] = v
    # value noise, amplitude +-6, on an 8-px lattice, nearest-upsampled then box-smoothed
    lat = rng.integers(-6, 7, (H // 8 + 2, W // 8 + 2)).astype(np.int16)
    noise = np.kron(lat, np.ones((8, 8), np.int16))[:H, :W]
    img = np.clip(img + noise + rng.integers(-2, 3, (H, W)), 0, 255)
    return img.astype(np.uint8)


def cut_frame(tex, width, height, t, cx, cy, margin=96):
    """Frame t of the sequence: translate (3t,0) and scale 1.01**t about (cx,cy); nearest sampling."""
    s = 1.01 ** t
    ys, xs = np.mgrid[0:height, 0:width].astype(np.float64)
    sx = (xs - cx) / s + cx - 3.0 * t + margin
    sy = (ys - cy) / s + cy + margin
    ix = np.clip(np.rint(sx).astype(np.int64), 0, tex.shape[1] - 1)
    iy = np.clip(np.rint(sy).astype(np.int64), 0, tex.shape[0] - 1)
    return np.ascontiguousarray(tex[iy, ix])


def disparity_map(width, height, seq=0):
    rng = _rng(seq, 998)
    bands = rng.integers(4, 65, 6)
    edges = np.linspace(0, width, 7).astype(int)
    d = np.zeros((height, width), np.int32)
    for b in range(6):
        d[:, edges[b]:edges[b + 1]] = bands[b]
    return d


def warp_right(left, disp):
    """Right image: right(x - d) = left(x); occlusions filled from the left neighbour."""
    h, w = left.shape
    right = np.zeros_like(left)
    filled = np.zeros((h, w), bool)
    xs = np.arange(w)
    for y in range(h):
        tx = xs - disp[y]
        ok = tx >= 0
        right[y, tx[ok]] = left[y, xs[ok]]
        filled[y, tx[ok]] = True
    # fill holes from the left neighbour (first column from the source)
    for x in range(w):
        hole = ~filled[:, x]
        if x == 0:
            right[hole, 0] = left[hole, 0]
        else:
            right[hole, x] = right[hole, x - 1]
    return right


def stereo_frame(seq=0, t=0, cfg=KITTI_STEREO, _cache={}):
    key = (seq, cfg["width"], cfg["height"])
    if key not in _cache:
        _cache.clear()
        _cache[key] = (base_texture(cfg["width"], cfg["height"], seq),
                       disparity_map(cfg["width"], cfg["height"], seq))
    tex, disp = _cache[key]
    left = cut_frame(tex, cfg["width"], cfg["height"], t, cfg["cx"], cfg["cy"])
    right = warp_right(left, disp)
    return left, right, 0.1 * t


def rgbd_frame(seq=0, t=0, cfg=KITTI03_RGBD, _cache={}):
    """(rgb HxWx3 u8, depth HxW u16, timestamp).  depth = round(DepthMapFactor * bf / disparity)."""
    key = (seq, cfg["width"], cfg["height"])
    if key not in _cache:
        _cache.clear()
        _cache[key] = (base_texture(cfg["width"], cfg["height"], seq),
                       disparity_map(cfg["width"], cfg["height"], seq))
    tex, disp = _cache[key]
    gray = cut_frame(tex, cfg["width"], cfg["height"], t, cfg["cx"], cfg["cy"])
    rng = _rng(seq, t)
    # 3 channels that are NOT equal, so the gray conversion is exercised
    dr = rng.integers(-3, 4, gray.shape); db = rng.integers(-3, 4, gray.shape)
    rgb = np.stack([np.clip(gray.astype(np.int16) + dr, 0, 255), gray.astype(np.int16),
                    np.clip(gray.astype(np.int16) + db, 0, 255)], axis=-1).astype(np.uint8)
    depth = np.rint(cfg["depth_map_factor"] * cfg["bf"] / disp).astype(np.uint16)
    return rgb, depth, (1.0 / cfg["fps"]) * t


def boxes_for_frame(seq, t, cfg=KITTI_STEREO, n_boxes=3):
    """Rows (id, cx, cy, w, h) as in yolov5_2Dbbox/%06d.txt (rgbd_my.cc:246-249).

    Box 0 moves with the background (static object), boxes 1.. move independently."""
    rng = _rng(seq, 997)
    W, H = cfg["width"], cfg["height"]
    rows = []
    for b in range(n_boxes):
        w = float(rng.integers(60, 221)); h = float(rng.integers(50, 161))
        cx0 = float(rng.integers(int(w), int(W - w))); cy0 = float(rng.integers(int(h / 2) + 1, int(H - h / 2)))
        vx = float(rng.choice([-4, 4])); vy = float(rng.choice([-1, 1]))
        if b == 0:
            s = 1.01 ** t
            # a texture point seen at x0 in frame 0 is seen at (x0 + 3t - cx) * s + cx in frame t
            cx = (cx0 + 3.0 * t - cfg["cx"]) * s + cfg["cx"]
            cy = (cy0 - cfg["cy"]) * s + cfg["cy"]
        else:
            cx, cy = cx0 + vx * t, cy0 + vy * t
        rows.append((b, cx, cy, w, h))
    return rows


def paste_objects(img, rows, seq):
    """Independently moving objects: every box but the first (which rides on the background) gets its own texture patch that moves
    rigidly with the box, so the geometric cull has something to reject.  In place on a gray image; returns it."""
    H, W = img.shape
    for (b, cx, cy, w, h) in rows:
        if b == 0:
            continue
        rng = np.random.Generator(np.random.PCG64(BASE_SEED + 77000 + 1000 * seq + int(b)))
        bw, bh = int(w), int(h)
        patch = np.full((bh, bw), 120, np.int16)
        n_rect = max(12, bw * bh // 120)
        xs = rng.integers(0, bw, n_rect); ys = rng.integers(0, bh, n_rect)
        ws = rng.integers(3, 24, n_rect); hs = rng.integers(3, 18, n_rect); vs = rng.integers(20, 236, n_rect)
        for x, y, ww, hh, v in zip(xs, ys, ws, hs, vs):
            patch[y:y + hh, x:x + ww] = v
        patch = np.clip(patch + rng.integers(-3, 4, patch.shape), 0, 255).astype(np.uint8)
        x0 = int(np.floor(cx - w / 2)); y0 = int(np.floor(cy - h / 2))
        xa, ya = max(x0, 0), max(y0, 0)
        xb, yb = min(x0 + bw, W), min(y0 + bh, H)
        if xb > xa and yb > ya:
            img[ya:yb, xa:xb] = patch[ya - y0:yb - y0, xa - x0:xb - x0]
    return img


def stereo_frame_dyn(seq=0, t=0, cfg=KITTI_STEREO, n_boxes=3):
    """stereo_frame with the moving objects of boxes_for_frame pasted into the left image before the disparity warp."""
    left, _, ts = stereo_frame(seq, t, cfg)
    left = paste_objects(left.copy(), boxes_for_frame(seq, t, cfg, n_boxes), seq)
    return left, warp_right(left, disparity_map(cfg["width"], cfg["height"], seq)), ts


def rgbd_frame_dyn(seq=0, t=0, cfg=KITTI03_RGBD, n_boxes=3):
    """rgbd_frame with the moving objects pasted into all three channels (the per-channel offsets are kept)."""
    rgb, depth, ts = rgbd_frame(seq, t, cfg)
    gray = paste_objects(rgb[:, :, 1].copy(), boxes_for_frame(seq, t, cfg, n_boxes), seq)
    changed = gray != rgb[:, :, 1]
    out = rgb.copy()
    for c in range(3):
        ch = out[:, :, c]
        ch[changed] = gray[changed]
    return out, depth, ts


def rows_to_rects(rows):
    """rgbd_my.cc:246-249: Rect2d(max(cx-w/2,0), max(cy-h/2,0), w, h) as (x, y, w, h) float64."""
    return np.array([[max(cx - w / 2, 0.0), max(cy - h / 2, 0.0), w, h] for (_, cx, cy, w, h) in rows], np.float64)


def mask_from_boxes(rows, width, height):
    """tools/mask.py:80-92 format: u8, 0 = background, 255 = instance; ellipse inscribed in each box."""
    m = np.zeros((height, width), np.uint8)
    yy, xx = np.mgrid[0:height, 0:width]
    for (_, cx, cy, w, h) in rows:
        m[((xx - cx) / (w / 2)) ** 2 + ((yy - cy) / (h / 2)) ** 2 <= 1.0] = 255
    return m


def random_image(width, height, seed, kind="texture"):
    """Small helper for unit tests: 'texture' (corner-rich), 'noise' (uniform random bytes) or 'mixed' (half of it low-contrast)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "noise":
        return rng.integers(0, 256, (height, width), dtype=np.uint8)
    tex = base_texture(width, height, seq=seed % 97, margin=0)
    tex = np.ascontiguousarray(tex[:height, :width])
    if kind == "mixed":            # left half at 1/8 contrast: cells whose corners only show at minThFAST (the extractor's retry), next to full-contrast cells
        half = width // 2
        tex[:, :half] = (100 + (tex[:, :half].astype(np.int32) - 128) // 8).astype(np.uint8)
    return tex


# ---------------------------------------------------------------------------------------------------------
# Synthetic ORB vocabulary in the shape of ORBvoc.txt (k = 10, L = 6, L1 scoring, TF-IDF weighting): the real
# file is a download that never was in the reference repository.
# ---------------------------------------------------------------------------------------------------------
def vocabulary(k=10, L=3, seed=7, early_leaf_frac=0.02, stop_frac=0.01):
    """Returns the content of the text file's node lines, in file order (parents before children):
    dict(k, L, scoring, weighting, parent int32[n], is_leaf u8[n], desc u8[n, 32], weight f64[n]); node id = line + 1."""
    rng = np.random.default_rng(seed)
    parent, is_leaf, desc, weight = [], [], [], []
    level_ids = np.array([0], np.int64)                      # node ids of the current level (root = 0)
    level_desc = rng.integers(0, 256, (1, 32), dtype=np.uint8)
    next_id = 1
    for level in range(1, L + 1):
        n_par = len(level_ids)
        par = np.repeat(level_ids, k)
        base = np.repeat(level_desc, k, axis=0)
        nflip = max(4, 96 >> level)                          # children drift less from their parent further down
        d = base.copy()
        bits = rng.integers(0, 256, (len(d), nflip))
        for j in range(nflip):
            d[np.arange(len(d)), bits[:, j] >> 3] ^= (1 << (bits[:, j] & 7)).astype(np.uint8)
        if level == 1:
            d = rng.integers(0, 256, (len(d), 32), dtype=np.uint8)
        leaf = np.ones(len(d), np.uint8) if level == L else (rng.random(len(d)) < early_leaf_frac).astype(np.uint8)
        w = np.where(leaf > 0, rng.uniform(0.5, 9.0, len(d)), 0.0)
        w = np.where((leaf > 0) & (rng.random(len(d)) < stop_frac), 0.0, w)
        ids = next_id + np.arange(len(d), dtype=np.int64)
        next_id += len(d)
        parent.append(par.astype(np.int32)); is_leaf.append(leaf); desc.append(d); weight.append(w.astype(np.float64))
        keep = leaf == 0
        level_ids, level_desc = ids[keep], d[keep]
        if len(level_ids) == 0:
            break
    return dict(k=k, L=L, scoring=0, weighting=0, parent=np.concatenate(parent), is_leaf=np.concatenate(is_leaf),
                desc=np.concatenate(desc), weight=np.concatenate(weight))


def write_vocabulary_text(voc, path):
    """TemplatedVocabulary::saveToTextFile's format: header `k L scoring weighting`, then one line per node:
    `parent isLeaf d0 .. d31 weight` (Thirdparty/DBoW2/DBoW2/TemplatedVocabulary.h:1427-1458)."""
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (voc["k"], voc["L"], voc["scoring"], voc["weighting"]))
        for i in range(len(voc["parent"])):
            f.write("%d %d %s %s\n" % (voc["parent"][i], voc["is_leaf"][i], " ".join(str(int(b)) for b in voc["desc"][i]),
                                       repr(float(voc["weight"][i]))))
