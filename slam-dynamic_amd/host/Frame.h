// Header-only mirror of the reference's FRAME-LEVEL class API over the C ABI (include/sd_frontend.h, sd_tracker_*):
//   ORB_SLAM2::Frame      include/Frame.h:46-238   the five constructors' results as the same public data members
//   ORB_SLAM2::Tracking   include/Tracking.h:65-70 GrabImageStereo / GrabImageRGBD / GrabImageMonocular (+ the overloads with
//                         boxes / mask), the dynamic block of Track_new (src/Tracking.cc:620-666) and q_frame / mLastFrame
//   ORB_SLAM2::System     include/System.h:66-79   TrackStereo / TrackRGBD / TrackMonocular with the reference's argument lists
// What is NOT here is the SLAM back end (pose tracking, local mapping, loop closing): System::Track* returns the pose the
// caller predicted (identity by default) -- a maintainer keeps their own System / Tracking and swaps the Frame construction
// and the dynamic block for this front end.
//
// Images, rectangles and key points are template parameters with the member names OpenCV uses, so the SAME code path serves
//   * cv::Mat / cv::Rect2d / cv::KeyPoint in a build that has OpenCV (nothing else is needed: cv::Mat has .data .cols .rows
//     .step .channels(), cv::Rect2d has .x .y .width .height, cv::KeyPoint is layout-compatible with sd_keypoint), and
//   * sdfe::Image / sdfe::Rect2d / sd_keypoint in this image, where OpenCV does not exist (tests/cpp/frame_mirror_main.cpp).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include "ORBextractor.h"
#include "Trajectory.h"

#define FRAME_GRID_ROWS 48      // include/Frame.h:39-40
#define FRAME_GRID_COLS 64

namespace sdfe {
struct Rect2d { double x = 0, y = 0, width = 0, height = 0; };
struct Point2d { double x = 0, y = 0; };
struct Image {                  // the subset of cv::Mat this path reads
    const uint8_t* data = nullptr;
    int cols = 0, rows = 0, nch = 1;
    size_t step = 0;            // bytes per row
    int elem = 1;               // bytes per channel element (2 for a CV_16U depth map, 4 for CV_32F)
    int channels() const { return nch; }
    size_t elemSize1() const { return (size_t)elem; }
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};
// a 4x4 CV_32F cv::Mat as Tracking uses it for mTcw / mVelocity; `valid == false` plays cv::Mat::empty()
struct Pose {
    float m[16]; bool valid = true;
    Pose() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.f : 0.f; }
    static Pose none() { Pose p; p.valid = false; return p; }
    bool empty() const { return !valid; }
};
// cv::Mat product of two 4x4 CV_32F matrices [OpenCV-recall: gemm accumulates CV_32F products in double, k ascending] -- mVelocity*mLastFrame.mTcw
inline Pose mul(const Pose& a, const Pose& b)
{
    Pose r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            double s = 0.0;
            for (int k = 0; k < 4; k++) s += (double)a.m[4 * i + k] * (double)b.m[4 * k + j];
            r.m[4 * i + j] = (float)s;
        }
    return r;
}
// Frame::UpdatePoseMatrices (src/Frame.cc:669-675): mRwc = mRcw.t(), mOw = -mRcw.t()*mtcw, returned as the 4x4 [mRwc | mOw]
inline Pose inverse_of(const Pose& Tcw)
{
    Pose r;
    for (int i = 0; i < 3; i++) {
        double s = 0.0;
        for (int k = 0; k < 3; k++) { r.m[4 * i + k] = Tcw.m[4 * k + i]; s += (double)Tcw.m[4 * k + i] * (double)Tcw.m[4 * k + 3]; }
        r.m[4 * i + 3] = -(float)s;
    }
    return r;
}
struct Settings {               // the YAML entries Tracking::Tracking reads (src/Tracking.cc:56-150)
    float fx = 0, fy = 0, cx = 0, cy = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0, k3 = 0, bf = 0, fps = 30, ThDepth = 40, DepthMapFactor = 1;
    int RGB = 1, width = 0, height = 0;
    int nFeatures = 1000, nLevels = 8, iniThFAST = 20, minThFAST = 7;
    float scaleFactor = 1.2f;
};
}  // namespace sdfe

namespace ORB_SLAM2 {

class Frame {
public:
    Frame() {}
    // Frame members (include/Frame.h:113-210), filled by Tracking::GrabImage* below
    double mTimeStamp = 0;
    float fx = 0, fy = 0, cx = 0, cy = 0, invfx = 0, invfy = 0, mbf = 0, mb = 0, mThDepth = 0;
    std::vector<sdfe::Rect2d> objects;
    std::vector<int> box_idx, box_status;          // box_status: 0 / 2 dynamic, -1 untouched (Tracking.cc:1093-1239)
    std::vector<sdfe::Point2d> box_velocity;
    std::vector<bool> omit;
    int N = 0, N_ori = 0, N_d = 0;
    std::vector<sd_keypoint> mvKeys, mvKeysUn;     // cv::KeyPoint layout
    std::vector<std::vector<sd_keypoint>> mvdynKeys, mvdynKeysUn;
    std::vector<float> mvuRight, mvDepth;
    std::vector<std::vector<float>> mvudynRight, mvdynDepth;
    std::vector<uint8_t> mDescriptors;             // N rows of 32 bytes (cv::Mat N x 32 CV_8U)
    std::vector<std::vector<uint8_t>> mdynDescriptors;
    std::vector<bool> mvbOutlier;
    float mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    std::vector<std::size_t> mGrid[FRAME_GRID_COLS][FRAME_GRID_ROWS];
    sdfe::Pose mTcw;
    long unsigned int mnId = 0;
    int mnScaleLevels = 0;
    float mfScaleFactor = 0, mfLogScaleFactor = 0;
    std::vector<float> mvScaleFactors, mvInvScaleFactors, mvLevelSigma2, mvInvLevelSigma2;
    float mnMinX = 0, mnMaxX = 0, mnMinY = 0, mnMaxY = 0;
    // results of Track_new's dynamic block for this frame
    int mnTrackHomoFlag = 0, mnSeparateRet = 0, mnRefFrameId = -1, mnTrackMatches = 0, mnLastMatches = -1;

    // Frame::GetFeaturesInArea (src/Frame.cc:735-788): key points of the grid cells a square window of half-width r touches, filtered by
    // |du| < r, |dv| < r and an optional octave range.  Cells are visited column by column, a cell's members in key-point order (the order the
    // matchers' tie-breaks depend on; the device walks the same order over its cell-sorted index list, k_proj_candidates).
    std::vector<size_t> GetFeaturesInArea(const float& x, const float& y, const float& r, const int minLevel = -1, const int maxLevel = -1) const
    {
        std::vector<size_t> hits;
        hits.reserve(N);
        // one axis of the window -> inclusive cell interval, empty (first > second) when the window misses the grid on that axis
        auto span = [r](float centre, float origin, float invCell, int nCells) {
            const int lo = (int)std::floor((centre - origin - r) * invCell), hi = (int)std::ceil((centre - origin + r) * invCell);
            if (lo >= nCells || hi < 0) return std::pair<int, int>(1, 0);
            return std::pair<int, int>(lo < 0 ? 0 : lo, hi > nCells - 1 ? nCells - 1 : hi);
        };
        const std::pair<int, int> cols = span(x, mnMinX, mfGridElementWidthInv, FRAME_GRID_COLS), rows = span(y, mnMinY, mfGridElementHeightInv, FRAME_GRID_ROWS);
        if (cols.first > cols.second || rows.first > rows.second) return hits;
        const bool levelFilter = minLevel > 0 || maxLevel >= 0;
        for (int c = cols.first; c <= cols.second; c++)
            for (int w = rows.first; w <= rows.second; w++)
                for (const std::size_t idx : mGrid[c][w]) {
                    const sd_keypoint& k = mvKeysUn[idx];
                    const bool levelOk = !levelFilter || (k.octave >= minLevel && (maxLevel < 0 || k.octave <= maxLevel));
                    if (levelOk && std::fabs(k.x - x) < r && std::fabs(k.y - y) < r) hits.push_back(idx);
                }
        return hits;
    }
    bool isInImage(const float& x, const float& y) const { return x >= mnMinX && x < mnMaxX && y >= mnMinY && y < mnMaxY; }   // include/Frame.h:93-96
};

// The per-frame front end of ORB_SLAM2::Tracking (one camera stream = one lane of an sd_tracker).
class Tracking {
public:
    enum { MONOCULAR = 0, STEREO = 1, RGBD = 2 };       // System::eSensor
    enum eTrackingState { AUTOMATIC = -2, SYSTEM_NOT_READY = -1, NO_IMAGES_YET = 0, NOT_INITIALIZED = 1, OK = 2, LOST = 3 };   // include/Tracking.h:92-98
    // channels: 1, 3 or 4 (Tracking.cc:175-200); depthIsFloat: the depth images are CV_32F instead of CV_16U (Tracking.cc:271-272)
    Tracking(const sdfe::Settings& s, int sensor, int channels, bool depthIsFloat = false)
        : mSensor(sensor), set_(s), ex_(s.nFeatures, s.scaleFactor, s.nLevels, s.iniThFAST, s.minThFAST), depthElem_(depthIsFloat ? 4 : 2)
    {
        sd_tracker_params p;
        std::memset(&p, 0, sizeof(p));
        p.sensor = sensor; p.width = s.width; p.height = s.height; p.channels = channels; p.rgb_order = s.RGB; p.n_lanes = 1; p.track_last = 1;
        p.depth_type = depthIsFloat ? SD_DEPTH_F32 : SD_DEPTH_U16;
        p.ini_features = sensor == MONOCULAR ? 2 * s.nFeatures : 0;       // mpIniORBextractor = new ORBextractor(2*nFeatures, ...) (Tracking.cc:127-128)
        p.cam.fx = s.fx; p.cam.fy = s.fy; p.cam.cx = s.cx; p.cam.cy = s.cy; p.cam.mbf = s.bf; p.cam.mb = s.bf / s.fx;
        const float K4[4] = {s.fx, s.fy, s.cx, s.cy}, d5[5] = {s.k1, s.k2, s.p1, s.p2, s.k3};
        float b4[4];
        sdfe::check(sd_image_bounds(s.width, s.height, K4, d5, b4), "ComputeImageBounds");      // Frame::ComputeImageBounds (Frame.cc:844-872)
        p.cam.mnMinX = b4[0]; p.cam.mnMaxX = b4[1]; p.cam.mnMinY = b4[2]; p.cam.mnMaxY = b4[3];
        for (int k = 0; k < 5; k++) p.dist[k] = d5[k];
        p.fps = s.fps; p.depth_map_factor = s.DepthMapFactor; p.th_depth = s.ThDepth;
        cam_ = p.cam;
        sdfe::check(sd_tracker_create(&trk_, ex_.handle(), &p), "Tracking");
        sdfe::check(sd_tracker_batch(trk_, &batch_), "Tracking");
    }
    ~Tracking() { if (trk_) sd_tracker_destroy(trk_); }
    Tracking(const Tracking&) = delete;
    Tracking& operator=(const Tracking&) = delete;

    // cv::Mat GrabImageStereo(imRectLeft, imRectRight, boxes, timestamp)   src/Tracking.cc:210-248
    template <class MatT, class RectT>
    sdfe::Pose GrabImageStereo(const MatT& imRectLeft, const MatT& imRectRight, std::vector<RectT>& boxes, const double& timestamp)
    {
        const uint8_t* im[2] = {(const uint8_t*)imRectLeft.data, (const uint8_t*)imRectRight.data};
        return grab(im, (size_t)imRectLeft.step, nullptr, 0, &boxes, timestamp);
    }
    // cv::Mat GrabImageStereo(imRectLeft, imRectRight, timestamp)          src/Tracking.cc:170-208
    template <class MatT>
    sdfe::Pose GrabImageStereo(const MatT& imRectLeft, const MatT& imRectRight, const double& timestamp)
    {
        const uint8_t* im[2] = {(const uint8_t*)imRectLeft.data, (const uint8_t*)imRectRight.data};
        return grab<sdfe::Rect2d>(im, (size_t)imRectLeft.step, nullptr, 0, nullptr, timestamp);
    }
    // cv::Mat GrabImageRGBD(imRGB, imD, mask, boxes, timestamp)            src/Tracking.cc:282-314.  The mask is accepted and not read:
    // Frame::firstSeparate never touches it (Frame.cc:555-604); its consumer is the dense mapper (sd_batch_backproject_dense).
    template <class MatT, class RectT>
    sdfe::Pose GrabImageRGBD(const MatT& imRGB, const MatT& imD, const MatT& /*mask*/, std::vector<RectT>& boxes, const double& timestamp)
    {
        const uint8_t* im[1] = {(const uint8_t*)imRGB.data};
        return grab(im, (size_t)imRGB.step, (const void*)imD.data, (size_t)imD.step / depthElem(imD), &boxes, timestamp);
    }
    // cv::Mat GrabImageRGBD(imRGB, imD, timestamp)                         src/Tracking.cc:251-280
    template <class MatT>
    sdfe::Pose GrabImageRGBD(const MatT& imRGB, const MatT& imD, const double& timestamp)
    {
        const uint8_t* im[1] = {(const uint8_t*)imRGB.data};
        return grab<sdfe::Rect2d>(im, (size_t)imRGB.step, (const void*)imD.data, (size_t)imD.step / depthElem(imD), nullptr, timestamp);
    }
    // cv::Mat GrabImageMonocular(im, timestamp)                            src/Tracking.cc:316-343
    template <class MatT>
    sdfe::Pose GrabImageMonocular(const MatT& im, const double& timestamp)
    {
        const uint8_t* p[1] = {(const uint8_t*)im.data};
        return grab<sdfe::Rect2d>(p, (size_t)im.step, nullptr, 0, nullptr, timestamp);
    }

    Frame mCurrentFrame, mLastFrame;
    int mSensor;
    // The SLAM state the pose side owns (include/Tracking.h:92-98,214-215), read at the next GrabImage*: mState (AUTOMATIC = the
    // sharded batch rule of DESIGN.md Q14) and mVelocity (empty until the motion model has one).  TrackHomo predicts the pose of the
    // new frame with `mCurrentFrame.SetPose(mVelocity*mLastFrame.mTcw)` (Tracking.cc:982); the pose side then overwrites
    // mCurrentFrame.mTcw with its estimate before the next frame arrives.
    int mState = AUTOMATIC;
    sdfe::Pose mVelocity = sdfe::Pose::none();
    // mCurrentFrame.mvpMapPoints as the pose side left them (TrackWithMotionModel / TrackLocalMap): world positions + flags (bit0: the
    // point exists and is not an outlier, bit1: Observations() > 0) of key points [0, n).  They become the points later frames project
    // from mLastFrame / q_frame (Tracking.cc:998-1010, 1714-1741) instead of the frame's own stereo points.
    void CommitMapPoints(const float* xw, const uint8_t* flags, int n)
    {
        int cap = 0;
        sd_batch_kp_capacity(batch_, &cap);
        if (n > cap) throw std::runtime_error("more map points than key points");
        std::vector<float> x((size_t)cap * 3, 0.f); std::vector<uint8_t> f((size_t)cap, 0);
        std::memcpy(x.data(), xw, (size_t)n * 12); std::memcpy(f.data(), flags, (size_t)n);
        const int32_t nn = n;
        sdfe::check(sd_tracker_set_mappoints(trk_, x.data(), f.data(), &nn), "Tracking::CommitMapPoints");
    }
    // The pose side's estimate of the frame just tracked (after TrackWithMotionModel / TrackLocalMap): replaces the prediction everywhere the
    // mirror keeps it -- mCurrentFrame.mTcw (the next prediction starts from it) and the trajectory record.
    void SetCurrentPose(const sdfe::Pose& Tcw)
    {
        mCurrentFrame.mTcw = Tcw;
        if (!mlFramePoses.empty()) std::memcpy(mlFramePoses.back().Tcw, Tcw.m, 64);
    }
    std::vector<sdfe::TrajectoryPose> mlFramePoses;
    sd_batch* batch() { return batch_; }
    int current_slot() const { return res_.cur_slot; }

private:
    template <class MatT> size_t depthElem(const MatT& imD) const
    {
        if ((size_t)imD.elemSize1() != depthElem_) throw std::runtime_error("depth image type differs from the one the tracker was created for");
        return depthElem_;
    }
    template <class RectT>
    sdfe::Pose grab(const uint8_t* const* images, size_t stride, const void* depth, size_t depthStrideElems, std::vector<RectT>* boxes,
                    const double& timestamp)
    {
        double bx[SD_MAX_BOXES][4];
        int32_t nb = -1;
        if (boxes) {
            if (boxes->size() > SD_MAX_BOXES) throw std::runtime_error("more than SD_MAX_BOXES boxes in a frame");
            nb = (int32_t)boxes->size();
            for (int j = 0; j < nb; j++) { bx[j][0] = (*boxes)[j].x; bx[j][1] = (*boxes)[j].y; bx[j][2] = (*boxes)[j].width; bx[j][3] = (*boxes)[j].height; }
        }
        const void* dp[1] = {depth};
        // the pose prior: mVelocity * mLastFrame.mTcw when there is a velocity (Tracking.cc:982), the last pose otherwise
        const bool haveLast = frames_ > 0;                 // mCurrentFrame still is the previous frame here (the reference's mLastFrame)
        sdfe::Pose Tcw = haveLast ? mCurrentFrame.mTcw : sdfe::Pose();
        if (!mVelocity.empty() && haveLast) Tcw = sdfe::mul(mVelocity, mCurrentFrame.mTcw);
        const sdfe::Pose Twc = sdfe::inverse_of(Tcw);
        if (mState != AUTOMATIC) {
            const int32_t st = ((mState == OK || mState == LOST) ? 1 : 0) | ((mState == OK && !mVelocity.empty()) ? 2 : 0);
            sdfe::check(sd_tracker_set_state(trk_, &st), "Tracking::GrabImage");
        } else sdfe::check(sd_tracker_set_state(trk_, nullptr), "Tracking::GrabImage");
        sdfe::check(sd_tracker_track_host(trk_, images, stride, depth ? dp : nullptr, depthStrideElems, boxes ? &bx[0][0] : nullptr, boxes ? &nb : nullptr,
                                          &timestamp, Tcw.m, Twc.m, &res_), "Tracking::GrabImage");
        mLastFrame = mCurrentFrame;
        fill(mCurrentFrame, timestamp);
        mCurrentFrame.mTcw = Tcw;
        frames_++;
        // mlRelativeFramePoses / mlFrameTimes / mlbLost (src/Tracking.cc:568-582), with the frame pose itself in place of the pose relative to
        // a reference key frame: what System::SaveTrajectoryTUM / KITTI write
        sdfe::TrajectoryPose tp;
        std::memcpy(tp.Tcw, Tcw.m, 64); tp.timestamp = timestamp; tp.lost = mState == LOST;
        mlFramePoses.push_back(tp);
        if (boxes) {                                    // boxTrack / firstSeparate rewrite the caller's vector (they take it by reference)
            boxes->resize(mCurrentFrame.objects.size());
            for (size_t j = 0; j < boxes->size(); j++) {
                (*boxes)[j].x = mCurrentFrame.objects[j].x; (*boxes)[j].y = mCurrentFrame.objects[j].y;
                (*boxes)[j].width = mCurrentFrame.objects[j].width; (*boxes)[j].height = mCurrentFrame.objects[j].height;
            }
        }
        return Tcw;
    }

    void fill(Frame& F, double timestamp)
    {
        const sd_lane_result& R = res_;
        const int slot = R.cur_slot;
        int cap = 0;
        sd_batch_kp_capacity(batch_, &cap);
        F = Frame();
        F.mTimeStamp = timestamp; F.mnId = (long unsigned int)R.frame_id;
        F.fx = set_.fx; F.fy = set_.fy; F.cx = set_.cx; F.cy = set_.cy; F.invfx = 1.0f / set_.fx; F.invfy = 1.0f / set_.fy;
        F.mbf = set_.bf; F.mb = set_.bf / set_.fx; F.mThDepth = set_.ThDepth;
        F.mnMinX = cam_.mnMinX; F.mnMaxX = cam_.mnMaxX; F.mnMinY = cam_.mnMinY; F.mnMaxY = cam_.mnMaxY;
        F.mfGridElementWidthInv = (float)FRAME_GRID_COLS / (F.mnMaxX - F.mnMinX);
        F.mfGridElementHeightInv = (float)FRAME_GRID_ROWS / (F.mnMaxY - F.mnMinY);
        F.mnScaleLevels = ex_.GetLevels(); F.mfScaleFactor = ex_.GetScaleFactor(); F.mfLogScaleFactor = std::log(F.mfScaleFactor);
        F.mvScaleFactors = ex_.GetScaleFactors(); F.mvInvScaleFactors = ex_.GetInverseScaleFactors();
        F.mvLevelSigma2 = ex_.GetScaleSigmaSquares(); F.mvInvLevelSigma2 = ex_.GetInverseScaleSigmaSquares();
        // keypoints, descriptors, stereo coordinates
        F.mvKeys.resize(cap); F.mDescriptors.resize((size_t)cap * 32); F.mvuRight.resize(cap); F.mvDepth.resize(cap);
        int n = 0;
        sdfe::check(sd_batch_download(batch_, slot, F.mvKeys.data(), F.mDescriptors.data(), cap, &n, nullptr), "download");
        sdfe::check(sd_batch_download_rgbd(batch_, slot, F.mvuRight.data(), F.mvDepth.data(), cap), "download");
        F.N = n; F.mvKeys.resize(n); F.mDescriptors.resize((size_t)n * 32); F.mvuRight.resize(n); F.mvDepth.resize(n);
        F.mvKeysUn.resize(cap);                        // UndistortKeyPoints (Frame.cc:812-842): a copy when Camera.k1 == 0
        int nu = 0;
        sdfe::check(sd_batch_download_keys_un(batch_, slot, F.mvKeysUn.data(), cap, &nu), "download");
        F.mvKeysUn.resize(nu);
        F.mvbOutlier.assign(n, false);
        // boxes and the per-box dynamic sets
        int nb = 0, nAll = 0, nStatic = 0;
        double bxs[SD_MAX_BOXES][4];
        int32_t idx[SD_MAX_BOXES], st[SD_MAX_BOXES], kept[SD_MAX_BOXES], start[SD_MAX_BOXES + 1];
        std::vector<int32_t> items(2 * (size_t)cap);
        sdfe::check(sd_batch_download_boxes(batch_, slot, &nb, &bxs[0][0], idx, st, kept, start, items.data(), (int)items.size(), &nAll, &nStatic), "download_boxes");
        F.N_ori = nStatic; F.N_d = nAll - nStatic;
        std::vector<sd_keypoint> dk(cap); std::vector<uint8_t> dd((size_t)cap * 32); std::vector<float> du(cap), dz(cap);
        int nd = 0;
        sdfe::check(sd_batch_download_dynamic(batch_, slot, dk.data(), dd.data(), du.data(), dz.data(), cap, &nd), "download_dynamic");
        std::vector<sd_keypoint> dku(cap);
        sdfe::check(sd_batch_download_dynamic_keys_un(batch_, slot, dku.data(), cap, &nd), "download_dynamic");
        F.objects.resize(nb); F.box_idx.assign(idx, idx + nb); F.box_status.assign(st, st + nb); F.omit.resize(nb); F.box_velocity.resize(nb);
        F.mvdynKeys.resize(nb); F.mvdynKeysUn.resize(nb); F.mdynDescriptors.resize(nb); F.mvudynRight.resize(nb); F.mvdynDepth.resize(nb);
        for (int j = 0; j < nb; j++) {
            F.objects[j].x = bxs[j][0]; F.objects[j].y = bxs[j][1]; F.objects[j].width = bxs[j][2]; F.objects[j].height = bxs[j][3];
            F.omit[j] = R.omit[j] != 0; F.box_velocity[j].x = R.box_velocity[j][0]; F.box_velocity[j].y = R.box_velocity[j][1];
            for (int k = start[j]; k < start[j + 1]; k++) {
                const int i = items[k];
                F.mvdynKeys[j].push_back(dk[i]); F.mvdynKeysUn[j].push_back(dku[i]);
                F.mdynDescriptors[j].insert(F.mdynDescriptors[j].end(), dd.begin() + (size_t)i * 32, dd.begin() + (size_t)i * 32 + 32);
                F.mvudynRight[j].push_back(du[i]); F.mvdynDepth[j].push_back(dz[i]);
            }
        }
        // the grid (AssignFeaturesToGrid / UpdateFeaturesToGrid): per-cell index lists in keypoint order
        std::vector<int16_t> cell(cap);
        sdfe::check(sd_batch_download_grid(batch_, slot, cell.data(), cap), "download_grid");
        for (int i = 0; i < n; i++)
            if (cell[i] >= 0) F.mGrid[cell[i] / FRAME_GRID_ROWS][cell[i] % FRAME_GRID_ROWS].push_back((size_t)i);
        F.mnTrackHomoFlag = R.track_flag; F.mnSeparateRet = R.separate_ret; F.mnRefFrameId = R.ref_frame_id;
        F.mnTrackMatches = R.n_track_matches; F.mnLastMatches = R.n_last_matches;
    }

    sdfe::Settings set_;
    ORBextractor ex_;
    sd_tracker* trk_ = nullptr;
    sd_batch* batch_ = nullptr;
    sd_camera cam_;
    sd_lane_result res_;
    size_t depthElem_ = 2;
    long frames_ = 0;
};

// ORB_SLAM2::System's tracking entry points (include/System.h:66-79, src/System.cc:119-375) with the reference's argument lists.
class System {
public:
    enum eSensor { MONOCULAR = 0, STEREO = 1, RGBD = 2 };
    System(const sdfe::Settings& settings, const eSensor sensor, int channels = 3, bool depthIsFloat = false)
        : mSensor(sensor), mpTracker(new Tracking(settings, (int)sensor, channels, depthIsFloat)) {}
    ~System() { delete mpTracker; }
    System(const System&) = delete;
    System& operator=(const System&) = delete;
    template <class MatT> sdfe::Pose TrackStereo(const MatT& imLeft, const MatT& imRight, const double& timestamp)
    { require(STEREO, "TrackStereo"); return mpTracker->GrabImageStereo(imLeft, imRight, timestamp); }
    template <class MatT, class RectT> sdfe::Pose TrackStereo(const MatT& imLeft, const MatT& imRight, std::vector<RectT>& boxes, const double& timestamp)
    { require(STEREO, "TrackStereo"); return mpTracker->GrabImageStereo(imLeft, imRight, boxes, timestamp); }
    template <class MatT> sdfe::Pose TrackRGBD(const MatT& im, const MatT& depthmap, const double& timestamp)
    { require(RGBD, "TrackRGBD"); return mpTracker->GrabImageRGBD(im, depthmap, timestamp); }
    template <class MatT, class RectT> sdfe::Pose TrackRGBD(const MatT& im, const MatT& depthmap, const MatT& mask, std::vector<RectT>& boxes, const double& timestamp)
    { require(RGBD, "TrackRGBD"); return mpTracker->GrabImageRGBD(im, depthmap, mask, boxes, timestamp); }
    template <class MatT> sdfe::Pose TrackMonocular(const MatT& im, const double& timestamp)
    { require(MONOCULAR, "TrackMonocular"); return mpTracker->GrabImageMonocular(im, timestamp); }
    Tracking* GetTracker() { return mpTracker; }
    // void SaveTrajectoryTUM(const string& filename) / SaveTrajectoryKITTI (src/System.cc:434-490, 524-565): refused for the monocular sensor
    bool SaveTrajectoryTUM(const std::string& filename) const
    { return mSensor != MONOCULAR && sdfe::SaveTrajectoryTUM(filename, mpTracker->mlFramePoses); }
    bool SaveTrajectoryKITTI(const std::string& filename) const
    { return mSensor != MONOCULAR && sdfe::SaveTrajectoryKITTI(filename, mpTracker->mlFramePoses); }

private:
    void require(eSensor s, const char* what) const
    {   // the reference prints "ERROR: you called TrackStereo but input sensor was not set to STEREO." and exit(-1)s (System.cc:121-125)
        if (mSensor != s) throw std::runtime_error(std::string("ERROR: you called ") + what + " but the input sensor was set to another type.");
    }
    eSensor mSensor;
    Tracking* mpTracker;
};

}  // namespace ORB_SLAM2
