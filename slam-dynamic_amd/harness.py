"""The callers of the hot path, restated for this library: the dataset layouts and file grammars
of the reference's example drivers, their per-frame timing statistics, and the per-frame call sequence of
System::TrackRGBD -> Tracking::GrabImageRGBD -> Frame(gray, rgb, depth, mask, boxes, last_frame, ...) ->
Tracking::Track_new's dynamic block, issued against the C ABI.  Host code only; everything heavy runs in the library.

  Examples/RGB-D/rgbd_my.cc:86-131    main loop: imread rgb / depth / mask, mask.convertTo(CV_32F), steady-clock around TrackRGBD
  Examples/RGB-D/rgbd_my.cc:133-146   sort(vTimesTrack); median = vTimesTrack[nImages/2]; mean = sum / nImages
  Examples/RGB-D/rgbd_my.cc:196-253   LoadKITTIImages: times.txt, image_2/%06d.png, depth/%06d.png, mask/mask_%06d.png,
                                      yolov5_2Dbbox/%06d.txt with lines `id cx cy w h`
  Examples/Stereo/stereo_kitti.cc:173-208  LoadImages: times.txt, image_2/, image_3/
  src/Tracking.cc:620-666, 952-959    reference-frame queue, TrackHomo, Separate, UpdateFrame
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
    """One frame at a time through the library, in the order the reference's Tracking thread does it for an RGB-D frame with
    detector boxes.  There is no SLAM back end here: the predicted pose of TrackHomo is the identity and the reference frame's
    map points are its own stereo points (SURVEY §8e's "sharded batch mode").  Slot 0 = mCurrentFrame, slot 1 = mLastFrame,
    slots 2.. = the copies held by q_frame."""

    def __init__(self, fe, cfg, rgb_order=True):
        import torch
        self.fe, self.cfg, self.torch = fe, cfg, torch
        self.ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        self.max_q = int(np.ceil(0.3 * cfg["fps"])) + 1
        self.ring = self.max_q + 2
        self.batch = fe.Batch(self.ex, cfg["width"], cfg["height"], 2 + self.ring)
        self.cam = fe.make_camera(cfg)
        self.queue = fe.RefQueue()
        self.rgb_order = rgb_order
        self.frame_no = 0
        self.last = None
        W, H = cfg["width"], cfg["height"]
        self.d_rgb = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
        self.d_depth = torch.empty((H, W), dtype=torch.int16, device="cuda")
        self.depth_factor = float(np.float32(1.0) / np.float32(cfg.get("depth_map_factor", 1.0)))
        self.I = np.eye(4, dtype=np.float32)[None]

    def close(self):
        self.batch.close()

    def track_rgbd(self, im_rgb, im_depth_u16, mask_f32, boxes, timestamp):
        """System::TrackRGBD(im, depthmap, mask, boxes, timestamp).  The mask is accepted and never read, exactly as
        Frame::firstSeparate does (Frame.cc:555-604: only a commented-out print touches it)."""
        fe, b, cfg, torch = self.fe, self.batch, self.cfg, self.torch
        W, H = cfg["width"], cfg["height"]
        st = torch.cuda.current_stream().cuda_stream
        # Tracking::GrabImageRGBD (Tracking.cc:256-272): cvtColor + depth scaling; Frame ctor (Frame.cc:297-323): extract + RGB-D stereo
        self.d_rgb.copy_(torch.from_numpy(np.ascontiguousarray(im_rgb)), non_blocking=False)
        self.d_depth.copy_(torch.from_numpy(np.ascontiguousarray(im_depth_u16).view(np.int16)))
        b.extract_color_device(self.d_rgb.data_ptr(), W * 3, W * H * 3, 1, bool(self.rgb_order), st)
        b.rgbd_from_u16(self.d_depth.data_ptr(), W, W * H, 1, self.depth_factor, cfg["bf"], st)
        # Frame::boxTrack (Frame.cc:324) on the host, then firstSeparate + the static/dynamic split (:329-367)
        boxes = np.asarray(boxes, np.float64).reshape(-1, 4)
        if self.last is not None:
            lo, li, lm, lv = self.last["objects"], self.last["box_idx"], self.last["omit"], self.last["velocity"]
        else:
            lo, li, lm, lv = np.zeros((0, 4)), np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 2))
        bx, idx, omit, vel = fe.box_track(boxes, lo, li, lm, lv, W, H)
        b.first_separate([0], [bx], [idx], stream=st)
        b.assign_grid(1, self.cam, st)
        b.unproject(1, 1, self.cam, self.I, st)
        kept = b.download_boxes(0)
        has_boxes = kept["nb"] > 0
        out = dict(flag=0, separate_ret=None, ref_slot=-1, n_boxes=kept["nb"], n_static=int(b.counts(1)[0]))
        # Tracking::Track_new dynamic block (Tracking.cc:620-666)
        if self.frame_no > 0 and has_boxes and len(self.queue) > 0:
            while True:
                ref = self.queue.candidate(timestamp, True)
                if ref < 0:
                    break
                th = 15.0                                            # TrackHomo: 15 unless stereo (7); doubled when < 20 matches
                b.search_by_projection([0], [ref], self.I, self.I, self.cam, th, False, True, stream=st)
                _, _, nm = b.download_matches(0)
                if nm < 20:
                    b.search_by_projection([0], [ref], self.I, self.I, self.cam, 2 * th, False, True, stream=st)
                    _, _, nm = b.download_matches(0)
                flag = 0
                if nm >= 20:
                    b.estimate_motion(st)
                    mo = b.download_motion(0)
                    flag = mo["flag"]
                if flag != 0:
                    lastb = b.download_boxes(1)
                    b.separate([0], [ref], None, None, [lastb["box_idx"]], [lastb["box_status"]], stream=st)     # HorF / flag stay on the device
                    b.update_frame(True, st)                         # if(Separate(...) == 1) mCurrentFrame.UpdateFrame(dynStatus)
                    b.assign_grid(1, self.cam, st)
                    ret, ds, dyn, mt = b.download_separate(0)
                    out.update(flag=flag, separate_ret=ret, ref_slot=ref, n_h=mo["n_h"], n_f=mo["n_f"], matches=int(nm))
                    break
                if not self.queue.reject():
                    break
        after = b.download_boxes(0)
        out.update(box_idx=after["box_idx"].copy(), box_status=after["box_status"].copy(), n_keypoints=int(b.counts(1)[0]))
        # the frame becomes mLastFrame and joins q_frame (Tracking.cc:952-959; both are copies in the reference too)
        b.copy_frame(0, 1, st)
        slot = 2 + self.frame_no % self.ring
        b.copy_frame(0, slot, st)
        self.queue.push(timestamp, slot, has_boxes, int(cfg["fps"]))
        ko = kept["kept_orig"]
        self.last = dict(objects=kept["boxes"].copy(), box_idx=kept["box_idx"].copy(), omit=omit[ko].copy(), velocity=vel[ko].copy())
        self.frame_no += 1
        b.sync()
        return out


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
