"""Frame-level oracle: the per-frame call sequence of the reference, restated over the CPU oracle's leaf functions.

TEST INFRASTRUCTURE ONLY (same rule as oracle.py): imported by tests/, smoke() and bench.py's cpu_baseline leg.

What is restated (file:line relative to /root/reference):
  Tracking::GrabImageStereo / GrabImageRGBD / GrabImageMonocular   src/Tracking.cc:170-343  (cvtColor, depth scaling)
  Frame::Frame x5                                                   src/Frame.cc:66-126 (stereo), 129-237 (stereo + boxes),
                                                                    240-294 (RGB-D), 297-403 (RGB-D + mask + boxes), 406-461 (mono)
  Tracking::Track_new, dynamic block + q_frame                      src/Tracking.cc:586-666, 952-959
  Tracking::TrackHomo                                               src/Tracking.cc:968-1086
  Tracking::Separate / Frame::UpdateFrame                           src/Tracking.cc:1093-1239, src/Frame.cc:607-641

There is no SLAM back end behind this oracle ("sharded batch mode"): the predicted pose of TrackHomo is
the pose the caller hands in (identity by default), a frame's map points are its own stereo points
(Frame::UnprojectStereo of every keypoint with depth > 0), `mState == OK && !mVelocity.empty()` holds from the third
frame on (the first frame initialises, the second gives the first velocity).  Spec Q9: the stereo + boxes
ctor's split is the RGB-D ctor's algorithm (the reference leaves N_d uninitialised there, Frame.cc:166-173).
"""
import numpy as np

SENSOR_MONOCULAR, SENSOR_STEREO, SENSOR_RGBD = 0, 1, 2        # System::eSensor, include/System.h:60-64


def pose_mul(a, b):
    """cv::Mat product of two 4x4 CV_32F matrices, as `mVelocity*mLastFrame.mTcw` (Tracking.cc:982) [OpenCV-recall: gemm accumulates CV_32F
    products in double, k ascending, and narrows once]."""
    a = np.asarray(a, np.float32).reshape(4, 4); b = np.asarray(b, np.float32).reshape(4, 4)
    r = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            s = 0.0
            for k in range(4):
                s += float(a[i, k]) * float(b[k, j])
            r[i, j] = np.float32(s)
    return r


def pose_inverse(Tcw):
    """Frame::UpdatePoseMatrices (Frame.cc:669-675): mRwc = mRcw.t(), mOw = -mRcw.t()*mtcw, as the 4x4 [mRwc | mOw]."""
    T = np.asarray(Tcw, np.float32).reshape(4, 4)
    r = np.eye(4, dtype=np.float32)
    for i in range(3):
        s = 0.0
        for k in range(3):
            r[i, k] = T[k, i]
            s += float(T[k, i]) * float(T[k, 3])
        r[i, 3] = -np.float32(s)
    return r


class FrameState:
    """The members of ORB_SLAM2::Frame this path produces (include/Frame.h:113-210)."""

    def __init__(self):
        self.mnId = -1
        self.mTimeStamp = 0.0
        self.kp = None; self.desc = None; self.ur = None; self.dep = None      # mvKeys, mDescriptors, mvuRight, mvDepth  [N]
        self.kpUn = None; self.dyn_kpUn = None                                  # mvKeysUn, mvdynKeysUn (== kp / dyn_kp when Camera.k1 == 0)
        self.N = 0
        self.dyn_kp = None; self.dyn_desc = None; self.dyn_ur = None; self.dyn_dep = None   # the N_d keypoints inside boxes
        self.objects = np.zeros((0, 4)); self.box_idx = np.zeros(0, np.int32); self.box_status = np.zeros(0, np.int32)
        self.omit = np.zeros(0, np.uint8); self.velocity = np.zeros((0, 2))
        self.boxStart = np.zeros(1, np.int32); self.boxItems = np.zeros(0, np.int32)  # mvdynKeys[b][k] = dyn[boxItems[boxStart[b] + k]]
        self.N_s = 0; self.N_d = 0
        self.cells = None                                                           # PosInGrid of every keypoint (x * 48 + y or -1)
        self.xw = None; self.mp_flags = None                                        # map-point table (UnprojectStereo)
        # Track_new's dynamic block
        self.ref_id = -1; self.track_flag = 0; self.n_track_matches = -1; self.pairs = None; self.motion = None
        self.separate_ret = None; self.dynStart = None; self.dynStatus = None; self.sep_matches = None; self.appended = None
        self.last_match = None; self.n_last_matches = -1


class SequenceOracle:
    """One camera stream through Tracking's front end, frame by frame."""

    def __init__(self, orc, cfg, sensor, rgb_order=True, track_last=True, threads=1, ini_features=0):
        self.orc, self.cfg, self.sensor = orc, cfg, sensor
        self.threads = threads               # 2: left / right extraction in two threads, as Frame.cc:87-90 / 151-154
        mk = lambda nf=cfg["n_features"]: orc.Extractor(nf, cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
        self.exL = mk()
        self.exR = mk() if sensor == SENSOR_STEREO else None
        # mpIniORBextractor = new ORBextractor(2*nFeatures, ...) for the monocular sensor (Tracking.cc:127-128), used by
        # GrabImageMonocular while mState is NOT_INITIALIZED / NO_IMAGES_YET (Tracking.cc:335-338)
        self.exIni = mk(ini_features) if (sensor == SENSOR_MONOCULAR and ini_features and ini_features != cfg["n_features"]) else None
        fx = np.float32(cfg["fx"]); bf = np.float32(cfg["bf"])
        self.K4 = np.array([fx, cfg["fy"], cfg["cx"], cfg["cy"]], np.float32)
        self.dist5 = np.array([cfg.get(k, 0.0) for k in ("k1", "k2", "p1", "p2", "k3")], np.float32)
        b = orc.image_bounds(cfg["width"], cfg["height"], self.K4, self.dist5)        # Frame::ComputeImageBounds (Frame.cc:844-872)
        self.cam10 = np.array([fx, cfg["fy"], cfg["cx"], cfg["cy"], bf, np.float32(bf / fx), b[0], b[1], b[2], b[3]], np.float32)
        self.rgb_order = rgb_order
        self.track_last = track_last
        dmf = np.float32(cfg.get("depth_map_factor", 1.0))
        self.depth_factor = float(np.float32(1.0) if abs(dmf) < 1e-5 else np.float32(1.0) / dmf)   # Tracking.cc:141-146
        self.q_frame = []                    # Tracking::q_frame (Tracking.h:109), front = index 0
        self.mLastFrame = None
        self.mMaxFrames = cfg["fps"]         # Tracking.cc:93-98: mMaxFrames = fps
        self.n_frames = 0
        self.I = np.eye(4, dtype=np.float32)

    def _undistort(self, kp):
        """Frame::UndistortKeyPoints (Frame.cc:812-842): a copy when k1 == 0, cv::undistortPoints on the positions otherwise."""
        out = kp.copy()
        if self.dist5[0] != 0 and len(kp):
            u = self.orc.undistort_points(np.stack([kp["x"], kp["y"]], 1), self.K4, self.dist5)
            out["x"] = u[:, 0]; out["y"] = u[:, 1]
        return out

    # ------------------------------------------------------------------ Frame::Frame
    def _gray(self, im):
        if im.ndim == 3:              # 3 or 4 channels: CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY (Tracking.cc:175-200)
            return self.orc.cvt_gray(im, 1 if self.rgb_order else 0)
        return np.ascontiguousarray(im, np.uint8)

    def construct(self, im, im2, boxes, timestamp, initialised=True):
        """GrabImage* + the Frame ctor.  im2 = right image (stereo), u16 or f32 depth (RGB-D) or None (mono);
        boxes = (n, 4) f64 rows x, y, w, h, or None for the ctors without boxes."""
        orc, cfg = self.orc, self.cfg
        F = FrameState()
        F.mnId = self.n_frames
        F.mTimeStamp = float(timestamp)
        if self.sensor == SENSOR_STEREO and self.threads >= 2:
            import threading
            box = {}
            th = threading.Thread(target=lambda: box.update(r=self.exR(self._gray(im2))))
            th.start()
            kp, desc = self.exL(self._gray(im))
            th.join()
            kpR, descR = box["r"]
        else:
            ex = self.exIni if (self.exIni is not None and not initialised) else self.exL
            kp, desc = ex(self._gray(im))
            if self.sensor == SENSOR_STEREO:
                kpR, descR = self.exR(self._gray(im2))
        N = len(kp)
        if N == 0:                                            # `if(mvKeys.empty()) return;`: no boxTrack, no members
            F.kp, F.desc, F.ur, F.dep = kp, desc, np.zeros(0, np.float32), np.zeros(0, np.float32)
            F.dyn_kp, F.dyn_desc, F.dyn_ur, F.dyn_dep = kp[:0], desc[:0], np.zeros(0, np.float32), np.zeros(0, np.float32)
            F.kpUn, F.dyn_kpUn = kp[:0], kp[:0]
            F.cells = np.zeros(0, np.int32)
            return F
        # stereo association (per keypoint: its order relative to the split is immaterial)
        if self.sensor == SENSOR_STEREO:
            ur, dep, _, _ = orc.stereo_matches(self.exL, self.exR, kp, desc, kpR, descR, cfg["bf"], cfg["fx"])
        elif self.sensor == SENSOR_RGBD:
            # `if((fabs(mDepthMapFactor-1.0f)>1e-5) || imDepth.type()!=CV_32F) imDepth.convertTo(imDepth,CV_32F,mDepthMapFactor)` (Tracking.cc:271-272)
            if im2.dtype == np.float32:
                dep32 = im2 if not abs(np.float32(self.depth_factor) - np.float32(1.0)) > 1e-5 else (im2 * np.float32(self.depth_factor)).astype(np.float32)
            else:
                dep32 = orc.depth_to_f32(im2, self.depth_factor)
            ur, dep = orc.stereo_from_rgbd(kp, dep32, cfg["bf"])           # depth looked up at the DISTORTED position (Frame.cc:1062-1065)
            if self.dist5[0] != 0:                                         # mvuRight[i] = kpU.pt.x - mbf / d (Frame.cc:1069), f32
                kun = self._undistort(kp)
                ur = np.where(dep > 0, kun["x"] - np.float32(cfg["bf"]) / np.where(dep > 0, dep, np.float32(1)), np.float32(-1)).astype(np.float32)
        else:
            ur = np.full(N, -1, np.float32); dep = np.full(N, -1, np.float32)
        if boxes is not None and self.sensor != SENSOR_MONOCULAR:
            L = self.mLastFrame
            if L is not None:
                lo, li, lm, lv = L.objects, L.box_idx, L.omit, L.velocity
            else:
                lo, li, lm, lv = np.zeros((0, 4)), np.zeros(0, np.int32), np.zeros(0, np.uint8), np.zeros((0, 2))
            bx, idx, omit, vel = orc.box_track(np.asarray(boxes, np.float64).reshape(-1, 4), lo, li, lm, lv, cfg["width"], cfg["height"])
            r = orc.first_separate(kp, desc, bx, idx, omit, vel)
            perm, Ns, Nd = r["perm"], r["Ns"], r["Nd"]
            ur2, dep2 = ur[perm], dep[perm]
            F.kp, F.desc, F.ur, F.dep = r["kp"][:Ns].copy(), r["desc"][:Ns].copy(), ur2[:Ns].copy(), dep2[:Ns].copy()
            F.dyn_kp, F.dyn_desc, F.dyn_ur, F.dyn_dep = r["kp"][Ns:].copy(), r["desc"][Ns:].copy(), ur2[Ns:].copy(), dep2[Ns:].copy()
            F.objects, F.box_idx, F.omit, F.velocity = r["boxes"].copy(), r["box_idx"].copy(), r["omit"].copy(), r["velocity"].copy()
            F.box_status = np.full(len(F.objects), -1, np.int32)
            F.boxStart = r["boxStart"].copy(); F.boxItems = r["boxItems"] - Ns      # indices into the dynamic arrays
            F.N_s, F.N_d = Ns, Nd
        else:
            F.kp, F.desc, F.ur, F.dep = kp, desc, ur, dep
            F.dyn_kp = kp[:0]; F.dyn_desc = desc[:0]; F.dyn_ur = ur[:0]; F.dyn_dep = dep[:0]
            F.N_s, F.N_d = N, 0
        F.N = len(F.kp)
        F.kpUn, F.dyn_kpUn = self._undistort(F.kp), self._undistort(F.dyn_kp)
        F.cells = orc.grid_cells(F.kpUn, self.cam10)
        return F

    # ------------------------------------------------------------------ Tracking::TrackHomo
    def _track_homo(self, F, R, Tcw, Trw):
        orc = self.orc
        th = 7.0 if self.sensor == SENSOR_STEREO else 15.0
        mono = self.sensor == SENSOR_MONOCULAR
        args = (F.kpUn, F.desc, F.ur, R.kpUn, R.desc, R.xw, R.mp_flags, Tcw, Trw, self.cam10, self.exL.scale)
        match, pairs, nm = orc.search_by_projection(*args, th, mono, True)
        if nm < 20:
            match, pairs, nm = orc.search_by_projection(*args, 2 * th, mono, True)
        F.n_track_matches, F.pairs = nm, pairs
        if nm < 20:
            return 0
        pl = np.stack([R.kpUn["x"][pairs[:, 0]], R.kpUn["y"][pairs[:, 0]]], 1)         # LastFrame.mvKeysUn[i].pt (ORBmatcher.cc:505)
        pc = np.stack([F.kpUn["x"][pairs[:, 1]], F.kpUn["y"][pairs[:, 1]]], 1)
        F.motion = orc.estimate_motion(pl, pc)
        return F.motion["flag"]

    # ------------------------------------------------------------------ Tracking::Track_new
    def set_mappoints(self, xw, flags):
        """The back end's MapPoints of the frame just tracked (mvpMapPoints after TrackWithMotionModel / TrackLocalMap): they replace the
        stereo points of mLastFrame, which is the same object as the newest q_frame entry."""
        L = self.mLastFrame
        n = len(flags)
        L.xw = np.zeros((max(n, L.N), 3), np.float32); L.mp_flags = np.zeros(max(n, L.N), np.uint8)
        L.xw[:n] = np.asarray(xw, np.float32).reshape(n, 3); L.mp_flags[:n] = flags

    def track(self, im, im2, boxes, timestamp, Tcw=None, Twc=None, state=None):
        """-> FrameState of mCurrentFrame after the dynamic block.  Tcw / Twc: pose of the current frame and its inverse
        (mVelocity * mLastFrame.mTcw in the reference; identity when omitted); the reference frame's pose is the one it was tracked with.
        state: the caller's SLAM state, bit0 = initialised, bit1 = mState == OK && !mVelocity.empty(); None = the automatic rule of the
        sharded batch mode (frame 0: not initialised -- the monocular sensor also on frame 1; velocity from frame 2 on)."""
        orc = self.orc
        if state is None:
            first, have_velocity, mono_init = self.n_frames == 0, self.n_frames >= 2, self.n_frames >= 2
        else:
            first, have_velocity, mono_init = not (state & 1), bool(state & 2), bool(state & 1)
        F = self.construct(im, im2, boxes, timestamp, initialised=mono_init)
        F.Tcw = self.I if Tcw is None else np.asarray(Tcw, np.float32).reshape(4, 4)
        F.Twc = self.I if Twc is None else np.asarray(Twc, np.float32).reshape(4, 4)
        if first:
            self.q_frame = []                                   # Tracking.cc:600-605
        if not first and len(F.objects) > 0 and len(self.q_frame) > 0:                    # Tracking.cc:622
            while len(self.q_frame) > 0 and F.mTimeStamp - self.q_frame[0].mTimeStamp > np.float32(0.2):
                R = self.q_frame[0]
                if len(R.objects) == 0:
                    self.q_frame.pop(0)
                    continue
                flag = self._track_homo(F, R, F.Tcw, R.Tcw) if have_velocity else 0
                F.ref_id = R.mnId
                if flag != 0:
                    F.track_flag = flag
                    cur = dict(kp=F.dyn_kpUn, desc=F.dyn_desc, boxStart=F.boxStart, boxItems=F.boxItems, box_idx=F.box_idx)      # mvdynKeysUn
                    ref = dict(kp=R.dyn_kpUn, desc=R.dyn_desc, boxStart=R.boxStart, boxItems=R.boxItems, box_idx=R.box_idx)
                    L = self.mLastFrame
                    ret, sc, ds, dyn, mt = orc.separate(F.motion["HorF"], flag, cur, ref, L.box_idx, L.box_status, F.box_status)
                    F.separate_ret, F.box_status, F.dynStart, F.dynStatus, F.sep_matches = ret, sc, ds, dyn, mt
                    if ret == 1:                                # mCurrentFrame.UpdateFrame(dynStatus)
                        app = orc.update_frame(F.dyn_kp, F.boxStart, F.boxItems, ds, dyn)
                        F.appended = app
                        F.kp = np.concatenate([F.kp, F.dyn_kp[app]]); F.desc = np.concatenate([F.desc, F.dyn_desc[app]])
                        F.ur = np.concatenate([F.ur, F.dyn_ur[app]]); F.dep = np.concatenate([F.dep, F.dyn_dep[app]])
                        F.N = len(F.kp)
                        F.kpUn = np.concatenate([F.kpUn, F.dyn_kpUn[app]])
                        F.cells = orc.grid_cells(F.kpUn, self.cam10)                # UpdateFeaturesToGrid
                    break
                if len(self.q_frame) == 1:
                    break
                self.q_frame.pop(0)
        # the frame's map points in sharded batch mode: its own stereo points
        Twc = F.Twc
        if self.sensor == SENSOR_MONOCULAR or F.N == 0:
            F.xw = np.zeros((F.N, 3), np.float32); F.mp_flags = np.zeros(F.N, np.uint8)
        else:
            F.xw, F.mp_flags = orc.unproject(F.kpUn, F.dep, self.cam10, Twc)
        # TrackWithMotionModel's matcher against mLastFrame (Tracking.cc:1714-1741): th 7 stereo / 15 otherwise
        if self.track_last and self.mLastFrame is not None and F.N > 0 and self.mLastFrame.N > 0:
            L = self.mLastFrame
            th = 7.0 if self.sensor == SENSOR_STEREO else 15.0
            F.last_match, _, F.n_last_matches = orc.search_by_projection(F.kpUn, F.desc, F.ur, L.kpUn, L.desc, L.xw, L.mp_flags, F.Tcw, L.Tcw,
                                                                         self.cam10, self.exL.scale, th, self.sensor == SENSOR_MONOCULAR, True)
        # mState == OK: queue + mLastFrame (Tracking.cc:952-959; both are copies)
        if len(self.q_frame) >= self.mMaxFrames * 0.3:
            self.q_frame.pop(0)
        self.q_frame.append(F)
        self.mLastFrame = F
        self.n_frames += 1
        return F
