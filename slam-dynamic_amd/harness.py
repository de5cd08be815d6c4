"""The callers of the hot path, restated for this library: the dataset layouts and file grammars
of the reference's example drivers, their per-frame timing statistics, and the per-frame call sequence of
System::TrackRGBD -> Tracking::GrabImageRGBD -> Frame(gray, rgb, depth, mask, boxes, last_frame, ...) ->
Tracking::Track_new's dynamic block, which is ONE call of the Frame-level boundary (sd_tracker_track).  Host code only.

  Examples/RGB-D/rgbd_my.cc:86-131    main loop: imread rgb / depth / mask, mask.convertTo(CV_32F), steady-clock around TrackRGBD
  Examples/RGB-D/rgbd_my.cc:133-146   sort(vTimesTrack); median = vTimesTrack[nImages/2]; mean = sum / nImages
  Examples/RGB-D/rgbd_my.cc:196-253   LoadKITTIImages: times.txt, image_2/%06d.png, depth/%06d.png, mask/mask_%06d.png,
                                      yolov5_2Dbbox/%06d.txt with lines `id cx cy w h`
  Examples/Stereo/stereo_kitti.cc:173-208  LoadImages: times.txt, image_2/, image_3/
"""
import os
import time
import numpy as np


# ------------------------------------------------------------------------------------------------ file grammars
def parse_box_text(text):
    """yolov5_2Dbbox/%06d.txt (rgbd_my.cc:236-251): every non-empty line `id cx cy w h` ->
    cv::Rect2d(MAX(cx - w/2, 0), MAX(cy - h/2, 0), w, h).  Returns (n, 4) float64 rows x, y, w, h."""
    rects = []
    for line in text.split("\n"):
        if not line:
            continue
        vals = line.split()
        _id, cx, cy, w, h = [float(v) for v in vals[:5]]
        rects.append([max(cx - w / 2, 0.0), max(cy - h / 2, 0.0), w, h])
    return np.array(rects, np.float64).reshape(-1, 4)


def format_box_rows(rows):
    return "".join("%d %r %r %r %r\n" % (int(i), float(cx), float(cy), float(w), float(h)) for (i, cx, cy, w, h) in rows)


def load_times(path):
    """times.txt: one timestamp per non-empty line (rgbd_my.cc:200-214, stereo_kitti.cc:176-190)."""
    out = []
    with open(path) as f:
        for s in f.read().split("\n"):
            if s:
                out.append(float(s.split()[0]))
    return out


def kitti_rgbd_layout(root, n_frames):
    """File names of LoadKITTIImages (rgbd_my.cc:229-235) + the parsed boxes (a missing box file = no boxes for that frame)."""
    frames = []
    for i in range(n_frames):
        s = "%06d" % i
        boxes = np.zeros((0, 4), np.float64)
        p = os.path.join(root, "yolov5_2Dbbox", s + ".txt")
        if os.path.exists(p):
            with open(p) as f:
                boxes = parse_box_text(f.read())
        frames.append(dict(rgb=os.path.join(root, "image_2", s + ".png"), depth=os.path.join(root, "depth", s + ".png"),
                           mask=os.path.join(root, "mask", "mask_" + s + ".png"), boxes=boxes))
    return frames


def kitti_stereo_layout(root, n_frames):
    """stereo_kitti.cc:192-207."""
    return [dict(left=os.path.join(root, "image_2", "%06d.png" % i), right=os.path.join(root, "image_3", "%06d.png" % i)) for i in range(n_frames)]


def imread_unchanged(path):
    """cv::imread(path, CV_LOAD_IMAGE_UNCHANGED) for the PNG kinds the drivers read: 8-bit gray / RGB (returned in BGR
    channel order as OpenCV does) and 16-bit depth."""
    from PIL import Image
    im = Image.open(path)
    a = np.array(im)
    if a.ndim == 3:
        a = a[:, :, ::-1].copy()                # PIL decodes RGB, cv::imread delivers BGR
    if a.dtype not in (np.uint8, np.uint16):
        a = a.astype(np.uint16)                 # PIL's mode "I" for 16-bit PNGs
    return a


def mask_to_f32(mask_u8):
    """mask.convertTo(mask, CV_32F) (rgbd_my.cc:101): 0 = background, non-zero = instance (tools/mask.py:80-92)."""
    return mask_u8.astype(np.float32)


def write_synthetic_kitti_rgbd(root, synth, seq, n_frames, cfg):
    """A synthetic sequence in the reference's on-disk layout (the real data sets are not available offline)."""
    from PIL import Image
    for d in ("image_2", "depth", "mask", "yolov5_2Dbbox"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    with open(os.path.join(root, "times.txt"), "w") as f:
        for t in range(n_frames):
            f.write("%e\n" % (t / cfg["fps"]))
    for t in range(n_frames):
        rgb, depth, _ = synth.rgbd_frame(seq, t, cfg)
        rows = synth.boxes_for_frame(seq, t, cfg)
        s = "%06d" % t
        Image.fromarray(rgb).save(os.path.join(root, "image_2", s + ".png"))
        Image.fromarray(depth.astype(np.uint16)).save(os.path.join(root, "depth", s + ".png"))
        Image.fromarray(synth.mask_from_boxes(rows, cfg["width"], cfg["height"])).save(os.path.join(root, "mask", "mask_" + s + ".png"))
        with open(os.path.join(root, "yolov5_2Dbbox", s + ".txt"), "w") as f:
            f.write(format_box_rows(rows))


def timing_summary(times):
    """rgbd_my.cc:133-146 / stereo_kitti.cc:157-170: sort, median = v[n/2], mean = total / n."""
    v = sorted(float(t) for t in times)
    n = len(v)
    return dict(median=v[n // 2], mean=sum(v) / n, n=n)


# ------------------------------------------------------------------------------------------------ the per-frame caller
class DynamicFrontEnd:
    """`System::TrackRGBD(im, depthmap, mask, boxes, timestamp)` for one camera stream, one frame at a time, as rgbd_my.cc's main loop
    calls it: a host image goes up, the Frame-level boundary (sd_tracker: one lane) does the rest -- GrabImageRGBD, the Frame constructor,
    Track_new's dynamic block, q_frame / mLastFrame.  There is no SLAM back end here: the tracker runs its sharded batch mode."""

    def __init__(self, fe, cfg, rgb_order=True):
        import torch
        self.fe, self.cfg, self.torch = fe, cfg, torch
        self.ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        self.trk = fe.Tracker(self.ex, cfg, fe.SENSOR_RGBD, 1, channels=3, rgb_order=rgb_order, track_last=True)
        W, H = cfg["width"], cfg["height"]
        self.d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        self.d_depth = torch.empty((H, W), dtype=torch.int16, device="cuda")

    def close(self):
        self.trk.close()

    def track_rgbd(self, im_rgb, im_depth_u16, mask_f32, boxes, timestamp):
        """The mask is accepted and never read, exactly as Frame::firstSeparate does (Frame.cc:555-604: only a commented-out print touches it)."""
        torch, cfg = self.torch, self.cfg
        W, H = cfg["width"], cfg["height"]
        self.d_rgb.copy_(torch.from_numpy(np.ascontiguousarray(im_rgb)))
        self.d_depth.copy_(torch.from_numpy(np.ascontiguousarray(im_depth_u16).view(np.int16)))
        R = self.trk.track(self.d_rgb.data_ptr(), W * 3, W * H * 3, [timestamp], boxes=[np.asarray(boxes, np.float64).reshape(-1, 4)],
                           d_depth=self.d_depth.data_ptr(), depth_stride=W, depth_pitch=W * H)[0]
        nb = R.n_boxes
        return dict(flag=R.track_flag, separate_ret=R.separate_ret if R.track_flag else None, ref_frame=R.ref_frame_id, n_boxes=nb, n_static=R.N_s,
                    n_keypoints=R.N, matches=R.n_track_matches, n_h=R.n_h, n_f=R.n_f, box_idx=np.array(R.box_idx[:nb], np.int32),
                    box_status=np.array(R.box_status[:nb], np.int32))


def run_rgbd_sequence(front, root, n_frames):
    """The main loop of rgbd_my.cc:86-131 (without the real-time sleep): returns per-frame results and the timing summary."""
    times = load_times(os.path.join(root, "times.txt"))
    frames = kitti_rgbd_layout(root, n_frames)
    results, v_times = [], []
    for ni, fr in enumerate(frames):
        im = imread_unchanged(fr["rgb"]); dm = imread_unchanged(fr["depth"]); mk = imread_unchanged(fr["mask"])
        if im.size == 0 or dm.size == 0 or mk.size == 0:
            raise RuntimeError("Failed to load image at: " + fr["rgb"])
        mask = mask_to_f32(mk)
        t1 = time.perf_counter()
        results.append(front.track_rgbd(im, dm, mask, fr["boxes"], times[ni]))
        v_times.append(time.perf_counter() - t1)
    return results, timing_summary(v_times)
