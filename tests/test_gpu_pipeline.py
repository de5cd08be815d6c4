"""GPU parity of the WHOLE per-frame chain against the frame-level oracle (oracle/pipeline.py), frame by frame and bit for bit:
Tracking::GrabImage* -> Frame::Frame -> boxTrack -> firstSeparate -> TrackHomo (SearchByProjection, th / 2*th retry, H / F fit) ->
Separate -> UpdateFrame -> grid -> match vs mLastFrame -> q_frame, through the Frame-level C ABI (sd_tracker_track).

  test_stereo_chain_kitti    BASELINE configs[2]: KITTI stereo 1241x376, 2000 features, boxes per frame (detector output), cull
  test_rgbd_chain_tum3       BASELINE configs[3]: TUM3 640x480 RGB-D, DepthMapFactor 5000, 30 fps, mask + boxes
  test_mono_chain_tum3       BASELINE configs[0]'s constructor (Frame.cc:406-461) through the same boundary
Lanes are chosen to hit every branch of Track_new's loop (Tracking.cc:620-666): frames without boxes in the queue, the constructor
without boxes, a reference frame that cannot be matched (flag 0 -> pop -> next candidate -> extra round), H and F outcomes.
"""
import numpy as np
import pytest

import __graft_entry__ as graft

pytestmark = pytest.mark.gpu


def _pipe():
    import importlib.util, os, sys
    if "sd_oracle_pipeline" in sys.modules:
        return sys.modules["sd_oracle_pipeline"]
    spec = importlib.util.spec_from_file_location("sd_oracle_pipeline", os.path.join(graft.ROOT, "oracle", "pipeline.py"))
    m = importlib.util.module_from_spec(spec); sys.modules["sd_oracle_pipeline"] = m; spec.loader.exec_module(m)
    return m


def _u32(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _check_frame(fe, trk, lane, R, F, tag):
    """R: sd_lane_result of the lane, F: oracle FrameState."""
    b = trk.batch
    S = trk.n_lanes
    slot = R.cur_slot
    assert R.frame_id == F.mnId, tag
    assert (R.N, R.N_s, R.N_d) == (F.N, F.N_s, F.N_d), "%s: N %r vs %r" % (tag, (R.N, R.N_s, R.N_d), (F.N, F.N_s, F.N_d))
    nb = len(F.objects)
    assert R.n_boxes == nb, "%s: boxes %d vs %d" % (tag, R.n_boxes, nb)
    assert np.array_equal(np.array(R.box_idx[:nb]), F.box_idx) and np.array_equal(np.array(R.box_status[:nb]), F.box_status), \
        "%s: box_idx/status %r %r vs %r %r" % (tag, R.box_idx[:nb], R.box_status[:nb], F.box_idx, F.box_status)
    assert np.array_equal(np.array([list(o) for o in R.objects[:nb]], np.float64).reshape(nb, 4), F.objects.reshape(nb, 4)), tag + ": objects"
    assert np.array_equal(np.array(R.omit[:nb], np.uint8), F.omit[:nb]), tag + ": omit"
    assert np.array_equal(np.array([list(v) for v in R.box_velocity[:nb]], np.float64).reshape(nb, 2), F.velocity.reshape(nb, 2)), tag + ": velocity"
    # Frame members
    kp, desc, _ = b.download(slot)
    assert kp.tobytes() == F.kp.tobytes(), tag + ": mvKeys"
    assert b.download_keys_un(slot).tobytes() == F.kpUn.tobytes(), tag + ": mvKeysUn"
    assert np.array_equal(desc, F.desc), tag + ": mDescriptors"
    ur, dep = b.download_rgbd(slot)
    assert np.array_equal(_u32(ur[:F.N]), _u32(F.ur)) and np.array_equal(_u32(dep[:F.N]), _u32(F.dep)), tag + ": mvuRight / mvDepth"
    g = b.download_boxes(slot)
    assert g["n_static"] + 0 == F.N_s or F.appended is not None, tag
    assert np.array_equal(g["boxStart"], F.boxStart) and np.array_equal(g["boxItems"], F.boxItems), tag + ": per-box lists"
    dk, dd, dur, ddep = b.download_dynamic(slot)
    assert dk.tobytes() == F.dyn_kp.tobytes() and np.array_equal(dd, F.dyn_desc), tag + ": mvdynKeys / mdynDescriptors"
    assert np.array_equal(_u32(dur), _u32(F.dyn_ur)) and np.array_equal(_u32(ddep), _u32(F.dyn_dep)), tag + ": mvudynRight / mvdynDepth"
    assert np.array_equal(b.download_grid(slot)[:F.N].astype(np.int32), F.cells), tag + ": grid"
    xw, fl = b.download_mappoints(slot)
    assert np.array_equal(fl[:F.N], F.mp_flags) and np.array_equal(_u32(xw[:F.N]), _u32(F.xw)), tag + ": map points"
    # Track_new's dynamic block
    assert R.ref_frame_id == F.ref_id, "%s: reference frame %d vs %d" % (tag, R.ref_frame_id, F.ref_id)
    assert R.track_flag == F.track_flag, "%s: TrackHomo flag %d vs %d" % (tag, R.track_flag, F.track_flag)
    if F.pairs is not None:                                   # TrackHomo's matcher ran
        assert R.n_track_matches == F.n_track_matches, "%s: nmatches %d vs %d" % (tag, R.n_track_matches, F.n_track_matches)
        _, pairs, nm = b.download_matches(lane)
        assert nm == F.n_track_matches and np.array_equal(pairs, F.pairs), tag + ": points_last / points_current"
    if F.motion is not None:
        mo = b.download_motion(lane)
        om = F.motion
        assert (mo["flag"], mo["n_h"], mo["n_f"]) == (om["flag"], om["n_h"], om["n_f"]), tag + ": H / F inliers"
        assert (R.n_h, R.n_f) == (om["n_h"], om["n_f"])
        assert np.array_equal(mo["mask_h"], om["mask_h"]) and np.array_equal(mo["mask_f"], om["mask_f"]), tag + ": inlier masks"
        for k in ("H", "F"):
            sc = max(1.0, float(np.abs(om[k]).max()))
            assert np.abs(mo[k] - om[k]).max() <= 1e-9 * sc, tag + ": " + k
        assert np.array_equal(_u32(mo["HorF"]), _u32(om["HorF"])), tag + ": HorF (f32) must be identical for classifyH / classifyF"
    if F.separate_ret is not None:
        assert R.separate_ret == F.separate_ret, "%s: Separate %d vs %d" % (tag, R.separate_ret, F.separate_ret)
        ret, ds, dyn, mt = b.download_separate(lane)
        assert ret == F.separate_ret and np.array_equal(ds[:nb + 1], F.dynStart) and np.array_equal(dyn, F.dynStatus) and np.array_equal(mt, F.sep_matches), \
            tag + ": dynStatus"
    else:
        assert R.separate_ret == 0
    if F.last_match is not None:
        assert R.n_last_matches == F.n_last_matches, "%s: matches vs mLastFrame %d vs %d" % (tag, R.n_last_matches, F.n_last_matches)
        m, _, nm = b.download_matches(S + lane)
        assert nm == F.n_last_matches and np.array_equal(m[:F.N], F.last_match), tag + ": mvpMapPoints after SearchByProjection(cur, last)"


def _run_chain(fe, orc, synth, cfg, sensor, lanes, n_frames, channels, depth_f32=False, ini_features=0):
    """lanes: list of dict(frames=callable t -> (im, im2), boxes=callable t -> (k,4) array or None, stamps=list) with the optional hooks of a
    caller that owns SLAM state: pose = callable t -> Tcw (4x4 f32; the pose prior mVelocity * mLastFrame.mTcw), state = callable t -> int
    (bit0 initialised, bit1 mState == OK && !mVelocity.empty()) -- all lanes or none --, mappoints = callable (t, FrameState) -> (xw, flags)
    or None: the MapPoints the pose side commits for the frame just tracked."""
    import torch
    P = _pipe()
    S = len(lanes)
    W, H = cfg["width"], cfg["height"]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    trk = fe.Tracker(ex, cfg, sensor, S, channels=channels, rgb_order=True, track_last=True, depth_f32=depth_f32, ini_features=ini_features)
    oracles = [P.SequenceOracle(orc, cfg, sensor, rgb_order=True, track_last=True, ini_features=ini_features) for _ in lanes]
    stats = dict(flag1=0, flag2=0, flag0=0, static=0, dynamic=0, status2=0, appended=0, refs=[], n_track=[], n_last=[], N=[], flags={})
    with_pose = "pose" in lanes[0]
    with_state = "state" in lanes[0]
    try:
        for t in range(n_frames):
            ims = [ln["frames"](t) for ln in lanes]
            bxs = [ln["boxes"](t) for ln in lanes]
            ts = [ln["stamps"][t] for ln in lanes]
            d_depth = None
            if sensor == fe.SENSOR_STEREO:
                d_img = torch.from_numpy(np.stack([np.stack([a, c]) for a, c in ims])).cuda()
                ipl = 2
            else:
                d_img = torch.from_numpy(np.stack([a for a, _ in ims])).cuda()
                ipl = 1
                if sensor == fe.SENSOR_RGBD:
                    dd = np.stack([c for _, c in ims])
                    d_depth = torch.from_numpy(dd if depth_f32 else dd.view(np.int16)).cuda()
            Tcw = Twc = None
            if with_pose:
                Tcw = np.stack([np.asarray(ln["pose"](t), np.float32).reshape(4, 4) for ln in lanes])
                Twc = np.stack([P.pose_inverse(T) for T in Tcw])
            states = [ln["state"](t) for ln in lanes] if with_state else [None] * S
            trk.set_state(states if with_state else None)
            res = trk.track(d_img.data_ptr(), W * channels, W * H * channels, ts, boxes=bxs if sensor != fe.SENSOR_MONOCULAR else None,
                            d_depth=d_depth.data_ptr() if d_depth is not None else 0, depth_stride=W, depth_pitch=W * H, Tcw=Tcw, Twc=Twc)
            commit_x, commit_f = [None] * S, [None] * S
            for l, ln in enumerate(lanes):
                F = oracles[l].track(ims[l][0], ims[l][1], bxs[l], ts[l], Tcw=Tcw[l] if with_pose else None, Twc=Twc[l] if with_pose else None, state=states[l])
                _check_frame(fe, trk, l, res[l], F, "frame %d lane %d" % (t, l))
                stats["n_track"].append(F.n_track_matches); stats["n_last"].append(F.n_last_matches); stats["N"].append(F.N)
                mp = ln["mappoints"](t, F) if "mappoints" in ln else None
                if mp is not None:
                    commit_x[l], commit_f[l] = mp
                    oracles[l].set_mappoints(mp[0], mp[1])
            if any(x is not None for x in commit_x):
                trk.set_mappoints(commit_x, commit_f)
            for l, ln in enumerate(lanes):
                F = oracles[l].mLastFrame
                stats["flag%d" % F.track_flag] += 1 if F.ref_id >= 0 else 0
                if F.ref_id >= 0: stats["refs"].append((t, l, F.ref_id))
                stats["flags"][(t, l)] = F.track_flag
                if F.separate_ret == 1: stats["static"] += 1
                if F.separate_ret == 0: stats["dynamic"] += 1
                stats["status2"] += int((F.box_status == 2).sum())
                stats["appended"] += 0 if F.appended is None else len(F.appended)
    finally:
        trk.close()
    return stats


def test_stereo_chain_kitti(gpu, fe, orc, synth):
    """BASELINE configs[2]: 7 consecutive synthetic KITTI stereo frames x 3 lanes."""
    cfg = synth.KITTI_STEREO
    T = 7
    stamps = [0.1 * t for t in range(T)]

    def lane(seq, boxes):
        return dict(frames=lambda t: synth.stereo_frame_dyn(seq, t, cfg)[:2], boxes=boxes, stamps=stamps)

    rect = lambda seq, t: synth.rows_to_rects(synth.boxes_for_frame(seq, t, cfg))
    lanes = [
        lane(21, lambda t: rect(21, t)),                                                   # boxes in every frame
        lane(22, lambda t: np.zeros((0, 4)) if t in (1, 2) else rect(22, t)),              # the detector finds nothing in frames 1, 2
        lane(23, lambda t: None),                                                          # Frame(imLeft, imRight, ...) without boxes
    ]
    st = _run_chain(fe, orc, synth, cfg, fe.SENSOR_STEREO, lanes, T, channels=1)
    assert st["flag1"] + st["flag2"] >= 5, "TrackHomo must succeed on the static-background lanes: %r" % st
    assert st["static"] >= 3 and st["appended"] > 0, "Separate must re-admit static boxes somewhere: %r" % st


def _yaw_pose(cfg, t, per_frame_px=3.0):
    """The pose prior of a camera that yaws so that the image moves `per_frame_px` px per frame (the synthetic sequences shift 3 px per frame and
    zoom by 1 %): Rcw = R_y(a t), a = per_frame_px / fx, no translation.  With it SearchByProjection's windows sit where the features went; with
    the identity they are up to 9 px (stereo, three frames back) off."""
    a = float(per_frame_px) * t / float(cfg["fx"])
    T = np.eye(4, dtype=np.float32)
    T[0, 0] = np.float32(np.cos(a)); T[0, 2] = np.float32(np.sin(a)); T[2, 0] = np.float32(-np.sin(a)); T[2, 2] = np.float32(np.cos(a))
    return T


def _commit_points(t, F, seed):
    """What a pose side would leave in mCurrentFrame.mvpMapPoints: most stereo points kept (slightly moved by its optimisation), some culled as
    outliers, some marked as observed by key frames, and a few key points WITHOUT depth given a map point (matched from the local map)."""
    rng = np.random.default_rng(1000 * seed + t)
    xw = F.xw.copy(); fl = F.mp_flags.copy()
    n = len(fl)
    xw += rng.normal(0, 0.01, xw.shape).astype(np.float32)
    fl[rng.random(n) < 0.15] = 0
    fl[(rng.random(n) < 0.4) & (fl != 0)] |= 2
    extra = (F.mp_flags == 0) & (rng.random(n) < 0.2)
    xw[extra] = rng.uniform(-20, 20, (int(extra.sum()), 3)).astype(np.float32) + np.array([0, 0, 30], np.float32)
    fl[extra] = 1
    return xw, fl


def test_stereo_chain_moving_camera_pose_state_mappoints(gpu, fe, orc, synth):
    """The SLAM state a live back end owns, through the Frame-level boundary (Tracking.cc:982 `SetPose(mVelocity*mLastFrame.mTcw)`, :998-1010 the
    reference frame's MapPoints, :971 `mState==OK && !mVelocity.empty()`): a yawing camera's pose prior per frame on both lanes; lane 0 with an
    explicit state that loses tracking on frame 5 (no TrackHomo there); lane 1 with the pose side's own MapPoints committed after every frame --
    every frame of both lanes bit for bit against the frame-level oracle fed the same state."""
    cfg = synth.KITTI_STEREO
    T = 8
    stamps = [0.1 * t for t in range(T)]
    rect = lambda seq, t: synth.rows_to_rects(synth.boxes_for_frame(seq, t, cfg))
    st0 = lambda t: 0 if t == 0 else (1 if t in (1, 5) else 3)
    st1 = lambda t: 0 if t == 0 else (1 if t == 1 else 3)
    lanes = [
        dict(frames=lambda t: synth.stereo_frame_dyn(24, t, cfg)[:2], boxes=lambda t: rect(24, t), stamps=stamps, pose=lambda t: _yaw_pose(cfg, t), state=st0),
        dict(frames=lambda t: synth.stereo_frame_dyn(25, t, cfg)[:2], boxes=lambda t: rect(25, t), stamps=stamps, pose=lambda t: _yaw_pose(cfg, t), state=st1,
             mappoints=lambda t, F: _commit_points(t, F, 25)),
    ]
    st = _run_chain(fe, orc, synth, cfg, fe.SENSOR_STEREO, lanes, T, channels=1)
    assert st["flags"][(5, 0)] == 0 and st["flags"][(5, 1)] != 0, "lane 0 is told tracking was lost on frame 5: no TrackHomo there: %r" % st["flags"]
    assert st["flag1"] + st["flag2"] >= 6, "TrackHomo must succeed under the pose prior: %r" % st
    assert max(v for v in st["n_track"] if v is not None) > 200, "the pose prior must put the windows on the features: %r" % st["n_track"]


def test_rgbd_chain_rgba_input_f32_depth(gpu, fe, orc, synth):
    """The inputs GrabImageRGBD accepts beside BGR + CV_16U (Tracking.cc:187-200 CV_RGBA2GRAY / CV_BGRA2GRAY, :271-272 a CV_32F depth map): 4-channel
    colour images and float depth, once in metres with DepthMapFactor 1 (passed through) and once still to be scaled (factor 5000), with the pose
    prior of a yawing camera -- bit for bit against the frame-level oracle."""
    T = 6
    for dmf in (1.0, 5000.0):
        cfg = dict(synth.TUM3); cfg["depth_map_factor"] = dmf
        stamps = [t / 30.0 for t in range(T)]

        def frames(seq):
            def f(t):
                rgb, depth, _ = synth.rgbd_frame_dyn(seq, t, synth.TUM3)
                rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 7 * t + 1, np.uint8)], -1)
                d = depth.astype(np.float32) if dmf != 1.0 else (depth.astype(np.float32) / np.float32(5000.0)).astype(np.float32)
                return np.ascontiguousarray(rgba), d
            return f

        rect = lambda seq, t: synth.rows_to_rects(synth.boxes_for_frame(seq, t, cfg))
        lanes = [dict(frames=frames(41), boxes=lambda t: rect(41, t), stamps=stamps, pose=lambda t: _yaw_pose(cfg, t)),
                 dict(frames=frames(42), boxes=lambda t: None if t == 1 else rect(42, t), stamps=stamps, pose=lambda t: _yaw_pose(cfg, t))]
        st = _run_chain(fe, orc, synth, cfg, fe.SENSOR_RGBD, lanes, T, channels=4, depth_f32=True)
        assert max(st["N"]) > 500 and max(v for v in st["n_last"] if v is not None) > 100, "%r" % st


def _sparse_rgbd(synth, cfg, k):
    """A nearly featureless frame: three rectangles on a flat background, all inside one box (so the frame has `objects`, but far
    too few map points for TrackHomo's 20 matches)."""
    W, H = cfg["width"], cfg["height"]
    g = np.full((H, W), 128, np.uint8)
    for j, (x, y) in enumerate(((200, 150), (260, 210), (330, 170))):
        g[y + 3 * k:y + 3 * k + 30, x:x + 40] = 40 + 60 * j
    rgb = np.stack([g, g, g], -1)
    depth = synth.rgbd_frame(34, 0, cfg)[1]
    return rgb, depth, np.array([[150.0, 100.0, 300.0, 200.0]])


def test_rgbd_chain_tum3(gpu, fe, orc, synth):
    """BASELINE configs[3]: TUM3 640x480 RGB-D (DepthMapFactor 5000, 30 fps), mask + boxes; lane 1 starts with three nearly
    featureless frames and has a gap in its time stamps, so that Track_new's loop rejects two reference frames (flag 0 -> pop)
    before it finds one it can match: the extra rounds of the tracker."""
    cfg = synth.TUM3
    T = 10
    st0 = [t / 30.0 for t in range(T)]
    st1 = [0.05 * t for t in range(6)] + [0.4 + 0.05 * (t - 6) for t in range(6, T)]

    def frames(seq):
        def f(t):
            rgb, depth, _ = synth.rgbd_frame_dyn(seq, t, cfg)
            return rgb, depth
        return f

    rect = lambda seq, t: synth.rows_to_rects(synth.boxes_for_frame(seq, t, cfg))
    f32 = frames(32)
    lanes = [
        dict(frames=frames(31), boxes=lambda t: rect(31, t), stamps=st0),
        dict(frames=lambda t: _sparse_rgbd(synth, cfg, t)[:2] if t < 3 else f32(t),
             boxes=lambda t: _sparse_rgbd(synth, cfg, t)[2] if t < 3 else rect(32, t), stamps=st1),
    ]
    st = _run_chain(fe, orc, synth, cfg, fe.SENSOR_RGBD, lanes, T, channels=3)
    assert st["flag0"] >= 1 and (5, 1, 0) in st["refs"], "lane 1 must see TrackHomo fail on a featureless frame: %r" % st
    assert (6, 1, 3) in st["refs"], "frame 6 of lane 1 must reject frames 1 and 2 and settle on frame 3 (two extra rounds): %r" % st
    assert st["flag1"] + st["flag2"] >= 4, "%r" % st


def test_rgbd_chain_tum1_lens_distortion(gpu, fe, orc, synth):
    """Examples/RGB-D/TUM1.yaml (Camera.k1 = 0.26: the shipped settings WITH lens distortion): mvKeysUn is a second key-point array, and
    the grid, both projection matchers, the RGB-D right coordinate, UnprojectStereo, TrackHomo's point pairs and classifyH / classifyF read it
    where the reference does, while box membership and the depth lookup stay on mvKeys -- 9 frames, two lanes, bit for bit."""
    cfg = synth.TUM1
    T = 9
    st = [t / 30.0 for t in range(T)]

    def frames(seq):
        def f(t):
            rgb, depth, _ = synth.rgbd_frame_dyn(seq, t, cfg)
            return rgb, depth
        return f

    rect = lambda seq, t: synth.rows_to_rects(synth.boxes_for_frame(seq, t, cfg))
    lanes = [dict(frames=frames(81), boxes=lambda t: rect(81, t), stamps=st),
             dict(frames=frames(82), boxes=lambda t: None if t == 2 else rect(82, t), stamps=st)]
    stt = _run_chain(fe, orc, synth, cfg, fe.SENSOR_RGBD, lanes, T, channels=3)
    assert stt["flag1"] + stt["flag2"] >= 3 and stt["appended"] > 0, "%r" % stt


def test_mono_chain_tum3(gpu, fe, orc, synth):
    cfg = dict(synth.TUM3)
    T = 3
    lanes = [dict(frames=lambda t: (synth.rgbd_frame(33, t, cfg)[0], None), boxes=lambda t: None, stamps=[t / 30.0 for t in range(T)])]
    _run_chain(fe, orc, synth, cfg, fe.SENSOR_MONOCULAR, lanes, T, channels=3)


def test_mono_chain_ini_extractor(gpu, fe, orc, synth):
    """GrabImageMonocular (Tracking.cc:316-343): mpIniORBextractor (2 * nFeatures, Tracking.cc:127-128) while the lane is not initialised, the
    regular extractor afterwards.  Lane 0 follows the automatic rule (frames 0, 1), lane 1 is told by its caller that initialisation took four
    frames; the second tracker run mixes the two kinds of lanes in one call."""
    cfg = dict(synth.TUM3)
    T = 6
    stamps = [t / 30.0 for t in range(T)]
    fr = lambda seq: (lambda t: (synth.rgbd_frame(seq, t, cfg)[0], None))
    st = _run_chain(fe, orc, synth, cfg, fe.SENSOR_MONOCULAR, [dict(frames=fr(35), boxes=lambda t: None, stamps=stamps)], 4, channels=3, ini_features=2 * cfg["n_features"])
    nf = cfg["n_features"]             # (the quadtree may return a few more than its quota)
    assert st["N"][0] > 1.5 * nf and st["N"][1] > 1.5 * nf and st["N"][2] < 1.2 * nf and st["N"][3] < 1.2 * nf, "frames 0, 1 come from the 2 * nFeatures extractor: %r" % st["N"]
    lanes = [dict(frames=fr(36), boxes=lambda t: None, stamps=stamps, state=lambda t: 0 if t < 2 else 3),
             dict(frames=fr(37), boxes=lambda t: None, stamps=stamps, state=lambda t: 0 if t < 4 else 3)]
    st = _run_chain(fe, orc, synth, cfg, fe.SENSOR_MONOCULAR, lanes, T, channels=3, ini_features=2 * cfg["n_features"])
    N = st["N"]          # [t0 l0, t0 l1, t1 l0, ...]
    assert N[2 * 3] < 1.2 * nf and N[2 * 3 + 1] > 1.5 * nf, "frame 3: lane 0 is initialised, lane 1 is not: %r" % N


@pytest.mark.parametrize("kind,seed,state", [("stereo", 11, False), ("rgbd", 12, False), ("stereo", 13, True), ("rgbd", 14, True)])
def test_randomised_sequences(gpu, kind, seed, state):
    """3 fixed draws per sensor of tools/fuzz_tracker.py: random box sets (none / empty / overlapping / partly outside / erased), jittered and
    jumping time stamps, blank frames (the constructor's `mvKeys.empty()` return), scene cuts -- every frame of every lane identical to the
    frame-level oracle.  (When written: 16 + 80 draws of 3 lanes x 8 frames were run; the sweep found the blank-frame case, where the
    reference's constructor returns before boxTrack and leaves `objects` empty.)"""
    import importlib.util, os
    spec = importlib.util.spec_from_file_location("fuzz_tracker", os.path.join(graft.ROOT, "tools", "fuzz_tracker.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    assert m.run(3, seed, kind, state=state) == 0        # state: + a caller's pose prior / mState / committed MapPoints and crowded (> 32 boxes) frames


@pytest.mark.parametrize("kind", ["stereo", "rgbd-tum1"])
def test_time_batched_prefetch_equals_sequential(gpu, fe, orc, synth, kind):
    """BASELINE configs[4]'s mode (whole sequences, few lanes per GPU): sd_tracker_prefetch runs the history-free half of D consecutive frames of
    every lane in ONE batch, the following D sd_tracker_track calls run only the recurrence (boxTrack -> firstSeparate -> TrackHomo -> Separate ->
    UpdateFrame -> match vs mLastFrame).  A stream cannot be cut into chunks with a halo -- boxTrack's ids (`max + 1`, Frame.cc:545-550) depend on
    its whole history -- so this is how a rank that owns one or two sequences fills the chip.  Two lanes, blocks of 4, 3 and 5 frames with two
    blocks outstanding: EVERY frame equals the sequential frame-level oracle bit for bit (no relabelling of box ids)."""
    import torch
    P = _pipe()
    stereo = kind == "stereo"
    cfg = synth.KITTI_STEREO if stereo else synth.TUM1
    sensor = fe.SENSOR_STEREO if stereo else fe.SENSOR_RGBD
    ch = 1 if stereo else 3
    W, H = cfg["width"], cfg["height"]
    S, blocks = 2, [4, 3, 5]
    T = sum(blocks)
    seqs = [91, 92]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    trk = fe.Tracker(ex, cfg, sensor, S, channels=ch, rgb_order=True, track_last=True, lookahead=5)
    oracles = [P.SequenceOracle(orc, cfg, sensor, rgb_order=True, track_last=True) for _ in range(S)]
    frames = [[(synth.stereo_frame_dyn(q, t, cfg)[:2] if stereo else synth.rgbd_frame_dyn(q, t, cfg)[:2]) for q in seqs] for t in range(T)]
    boxes = [[synth.rows_to_rects(synth.boxes_for_frame(q, t, cfg)) if not (l == 1 and t == 2) else None for l, q in enumerate(seqs)] for t in range(T)]
    stamp = lambda t: t / float(cfg["fps"])

    def upload(t0, n):
        if stereo:
            img = torch.from_numpy(np.stack([np.stack([np.stack(frames[t][l]) for l in range(S)]) for t in range(t0, t0 + n)])).cuda()      # [n, S, 2, H, W]
            return img, None
        img = torch.from_numpy(np.stack([np.stack([frames[t][l][0] for l in range(S)]) for t in range(t0, t0 + n)])).cuda()
        dep = torch.from_numpy(np.stack([np.stack([frames[t][l][1] for l in range(S)]) for t in range(t0, t0 + n)]).view(np.int16)).cuda()
        return img, dep

    flags = 0
    try:
        starts = np.cumsum([0] + blocks)
        keep = []
        pending = []
        # block 0 and block 1 are both in flight before the first frame is tracked
        for bi in (0, 1):
            img, dep = upload(starts[bi], blocks[bi]); keep.append((img, dep))
            trk.prefetch(img.data_ptr(), W * ch, W * H * ch, blocks[bi], d_depth=dep.data_ptr() if dep is not None else 0, depth_stride=W, depth_pitch=W * H)
            pending.append(bi)
        nxt = 2
        for t in range(T):
            if t in (1, 4):
                # a call that fails on its arguments (more than SD_MAX_BOXES boxes) must leave the pool of prefetched frames -- and the lanes -- as they
                # were: the frame it would have consumed is still the next one (mid-block at t = 1, a block's first frame at t = 4)
                with pytest.raises(fe.SdError):
                    trk.track(0, W * ch, W * H * ch, [stamp(t)] * S, boxes=np.zeros((S, fe.MAXB, 4)), n_boxes=np.full(S, fe.MAXB + 1, np.int32))
            res = trk.track(0, W * ch, W * H * ch, [stamp(t)] * S, boxes=boxes[t])
            for l in range(S):
                F = oracles[l].track(frames[t][l][0], frames[t][l][1], boxes[t][l], stamp(t))
                _check_frame(fe, trk, l, res[l], F, "%s frame %d lane %d" % (kind, t, l))
                flags += F.track_flag != 0
            if t + 1 == starts[pending[0] + 1]:          # a block is used up: the next one goes out while the other outstanding block is tracked
                pending.pop(0)
                if nxt < len(blocks):
                    img, dep = upload(starts[nxt], blocks[nxt]); keep.append((img, dep))
                    trk.prefetch(img.data_ptr(), W * ch, W * H * ch, blocks[nxt], d_depth=dep.data_ptr() if dep is not None else 0, depth_stride=W, depth_pitch=W * H)
                    pending.append(nxt); nxt += 1
        with pytest.raises(fe.SdError):
            trk.track(0, W * ch, W * H * ch, [stamp(T)] * S, boxes=boxes[0])           # nothing prefetched and no image
    finally:
        trk.close()
    assert flags >= 4, "TrackHomo must have run on the later frames"


@pytest.mark.parametrize("kind", ["stereo", "rgbd-tum1"])
def test_export_import_prefetched_between_trackers(gpu, fe, orc, synth, kind):
    """Frames sharded over GPUs (BASELINE configs[4]): the history-free half of a frame runs in a WORKER tracker (sd_tracker_prefetch), leaves it as
    fixed-stride records (sd_tracker_export_prefetched, here in two pieces with first_frame > 0 and a stride wider than the record: the caller's own
    bytes behind every record must survive), and enters the OWNER tracker as a prefetched block (sd_tracker_import_prefetched) whose frames
    sd_tracker_track(d_images = NULL) consumes.  Two lanes, blocks of 3, 4 and 3 frames, two blocks outstanding in the owner; a distorted RGB-D camera
    (TUM1: the record then carries mvKeysUn too) and a stereo one.  EVERY frame equals the sequential frame-level oracle bit for bit."""
    import torch
    P = _pipe()
    stereo = kind == "stereo"
    cfg = synth.KITTI_STEREO if stereo else synth.TUM1
    sensor = fe.SENSOR_STEREO if stereo else fe.SENSOR_RGBD
    ch = 1 if stereo else 3
    W, H = cfg["width"], cfg["height"]
    S, blocks = 2, [3, 4, 3]
    T = sum(blocks)
    seqs = [93, 94]
    ex = fe.ORBextractor(cfg["n_features"], cfg["scale_factor"], cfg["n_levels"], cfg["ini_th_fast"], cfg["min_th_fast"])
    owner = fe.Tracker(ex, cfg, sensor, S, channels=ch, rgb_order=True, track_last=True, lookahead=4)
    worker = fe.Tracker(ex, cfg, sensor, S, channels=ch, rgb_order=True, track_last=False, lookahead=4)
    plain = fe.Tracker(ex, cfg, sensor, S, channels=ch, rgb_order=True, track_last=True)
    oracles = [P.SequenceOracle(orc, cfg, sensor, rgb_order=True, track_last=True) for _ in range(S)]
    frames = [[(synth.stereo_frame_dyn(q, t, cfg)[:2] if stereo else synth.rgbd_frame_dyn(q, t, cfg)[:2]) for q in seqs] for t in range(T)]
    boxes = [[synth.rows_to_rects(synth.boxes_for_frame(q, t, cfg)) for q in seqs] for t in range(T)]
    stamp = lambda t: t / float(cfg["fps"])

    def upload(t0, n):
        if stereo:
            return torch.from_numpy(np.stack([np.stack([np.stack(frames[t][l]) for l in range(S)]) for t in range(t0, t0 + n)])).cuda(), None
        img = torch.from_numpy(np.stack([np.stack([frames[t][l][0] for l in range(S)]) for t in range(t0, t0 + n)])).cuda()
        dep = torch.from_numpy(np.stack([np.stack([frames[t][l][1] for l in range(S)]) for t in range(t0, t0 + n)]).view(np.int16)).cuda()
        return img, dep

    try:
        R = owner.record_bytes()
        assert R > 0 and R % 16 == 0 and R == worker.record_bytes() and plain.record_bytes() == 0
        TAIL = 48
        stride = R + TAIL
        st = torch.cuda.Stream()
        starts = np.cumsum([0] + blocks)
        keep, recs = [], []

        def hand_over(bi):
            n = blocks[bi]
            img, dep = upload(starts[bi], n); keep.append((img, dep))
            worker.prefetch(img.data_ptr(), W * ch, W * H * ch, n, d_depth=dep.data_ptr() if dep is not None else 0, depth_stride=W, depth_pitch=W * H, stream=st.cuda_stream)
            rec = torch.full((n * S, stride), 0xA5, dtype=torch.uint8, device="cuda"); recs.append(rec)
            n1 = max(1, n // 2)                               # two pieces: [0, n1) and [n1, n)
            worker.export_prefetched(0, n1, rec.data_ptr(), record_stride=stride, stream=st.cuda_stream)
            worker.export_prefetched(n1, n - n1, rec[n1 * S:].data_ptr(), record_stride=stride, stream=st.cuda_stream)
            worker.discard_prefetched()
            owner.import_prefetched(rec.data_ptr(), n, record_stride=stride, stream=st.cuda_stream)
            return rec

        with pytest.raises(fe.SdError):
            worker.export_prefetched(0, 1, torch.zeros(stride * S, dtype=torch.uint8, device="cuda").data_ptr(), record_stride=stride)      # nothing prefetched
        with pytest.raises(fe.SdError):
            owner.import_prefetched(torch.zeros(16, dtype=torch.uint8, device="cuda").data_ptr(), 5, record_stride=stride)                   # more than lookahead
        with pytest.raises(fe.SdError):
            owner.import_prefetched(torch.zeros(16, dtype=torch.uint8, device="cuda").data_ptr(), 1, record_stride=R - 16)                  # stride below the record
        hand_over(0); hand_over(1)                                # two blocks outstanding in the owner before the first frame is tracked
        nxt, pending, flags = 2, [0, 1], 0
        for t in range(T):
            res = owner.track(0, W * ch, W * H * ch, [stamp(t)] * S, boxes=boxes[t])
            for l in range(S):
                F = oracles[l].track(frames[t][l][0], frames[t][l][1], boxes[t][l], stamp(t))
                _check_frame(fe, owner, l, res[l], F, "%s frame %d lane %d" % (kind, t, l))
                flags += F.track_flag != 0
            if t + 1 == starts[pending[0] + 1]:
                pending.pop(0)
                if nxt < len(blocks):
                    hand_over(nxt); pending.append(nxt); nxt += 1
        torch.cuda.synchronize()
        for rec in recs:                                          # the bytes behind every record are the caller's
            assert bool((rec[:, R:] == 0xA5).all()), "export wrote past the record"
        assert flags >= 1, "TrackHomo must have run on the later frames"
    finally:
        owner.close(); worker.close(); plain.close()
