// Exercises slam-dynamic_amd/host/Frame.h (ORB_SLAM2::System / Tracking / Frame mirror) end to end on the GPU:
//   frame_mirror_main <stereo|rgbd> <w> <h> <channels> <n_frames> <in.bin> <out.bin> <fx> <fy> <cx> <cy> <bf> <fps> <DepthMapFactor> <nFeatures> <iniTh> [k1 k2 p1 p2 k3]
//                     [ext depthf32]   ext = 1: the caller owns SLAM state (below); depthf32 = 1: the depth maps are CV_32F
// in.bin, per frame: [ext: i32 mState, i32 has_velocity, 16 f32 mVelocity] f64 timestamp, i32 n_boxes (-1 = the overload without boxes), n_boxes x 4 f64,
// image 0 bytes, then image 1 bytes (stereo) or the depth map (rgbd), then (rgbd) the 8-bit mask [ext: i32 n_mp (-1 = none), n_mp x 3 f32 world positions,
// n_mp u8 flags = the map points the pose side commits for this frame AFTER it was tracked].
// kind "mono": one image per frame, TrackMonocular (mpIniORBextractor with 2 * nFeatures while the lane is not initialised).
// out.bin, per frame: the Frame members the parity test compares (see dump()).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "Frame.h"

template <class T> static void put(FILE* o, const T* p, size_t n) { fwrite(p, sizeof(T), n, o); }
static void put_i(FILE* o, int32_t v) { fwrite(&v, 4, 1, o); }

static void dump(FILE* o, const ORB_SLAM2::Frame& F)
{
    put_i(o, F.N); put_i(o, F.N_ori); put_i(o, F.N_d); put_i(o, (int32_t)F.objects.size());
    put_i(o, F.mnTrackHomoFlag); put_i(o, F.mnSeparateRet); put_i(o, F.mnRefFrameId); put_i(o, (int32_t)F.mnId);
    put(o, F.mvKeys.data(), F.mvKeys.size()); put(o, F.mvKeysUn.data(), F.mvKeysUn.size()); put(o, F.mDescriptors.data(), F.mDescriptors.size());
    put(o, F.mvuRight.data(), F.mvuRight.size()); put(o, F.mvDepth.data(), F.mvDepth.size());
    for (size_t j = 0; j < F.objects.size(); j++) {
        const double r[4] = {F.objects[j].x, F.objects[j].y, F.objects[j].width, F.objects[j].height};
        put(o, r, 4); put_i(o, F.box_idx[j]); put_i(o, F.box_status[j]); put_i(o, F.omit[j] ? 1 : 0);
        const double v[2] = {F.box_velocity[j].x, F.box_velocity[j].y};
        put(o, v, 2);
        put_i(o, (int32_t)F.mvdynKeys[j].size());
        put(o, F.mvdynKeys[j].data(), F.mvdynKeys[j].size()); put(o, F.mvdynKeysUn[j].data(), F.mvdynKeysUn[j].size()); put(o, F.mdynDescriptors[j].data(), F.mdynDescriptors[j].size());
        put(o, F.mvudynRight[j].data(), F.mvudynRight[j].size()); put(o, F.mvdynDepth[j].data(), F.mvdynDepth[j].size());
    }
    std::vector<int32_t> cell(F.N, -1);
    for (int ix = 0; ix < FRAME_GRID_COLS; ix++)
        for (int iy = 0; iy < FRAME_GRID_ROWS; iy++) {
            size_t last = 0;
            for (size_t k = 0; k < F.mGrid[ix][iy].size(); k++) {
                const size_t i = F.mGrid[ix][iy][k];
                if (k > 0 && i <= last) { fprintf(stderr, "grid cell not in keypoint order\n"); exit(9); }
                last = i; cell[i] = ix * FRAME_GRID_ROWS + iy;
            }
        }
    put(o, cell.data(), cell.size());
    put(o, F.mTcw.m, 16);
    // GetFeaturesInArea around the first keypoint must contain it
    if (F.N > 0) {
        const std::vector<size_t> v = F.GetFeaturesInArea(F.mvKeysUn[0].x, F.mvKeysUn[0].y, 10.f);
        bool found = false;
        for (size_t i : v) found |= i == 0;
        if (cell[0] >= 0 && !found) { fprintf(stderr, "GetFeaturesInArea lost the keypoint\n"); exit(8); }
    }
}

int main(int argc, char** argv)
{
    if (argc < 17) return 2;
    const std::string kind = argv[1];
    const int w = atoi(argv[2]), h = atoi(argv[3]), ch = atoi(argv[4]), nf = atoi(argv[5]);
    sdfe::Settings s;
    s.width = w; s.height = h; s.fx = (float)atof(argv[8]); s.fy = (float)atof(argv[9]); s.cx = (float)atof(argv[10]); s.cy = (float)atof(argv[11]);
    s.bf = (float)atof(argv[12]); s.fps = (float)atof(argv[13]); s.DepthMapFactor = (float)atof(argv[14]); s.nFeatures = atoi(argv[15]); s.iniThFAST = atoi(argv[16]);
    if (argc >= 22) { s.k1 = (float)atof(argv[17]); s.k2 = (float)atof(argv[18]); s.p1 = (float)atof(argv[19]); s.p2 = (float)atof(argv[20]); s.k3 = (float)atof(argv[21]); }
    const bool ext = argc >= 23 && atoi(argv[22]) != 0, depthf32 = argc >= 24 && atoi(argv[23]) != 0;
    FILE* in = fopen(argv[6], "rb"); FILE* out = fopen(argv[7], "wb");
    if (!in || !out) return 3;
    const bool stereo = kind == "stereo", mono = kind == "mono";
    const size_t depthElem = depthf32 ? 4 : 2;
    try {
        ORB_SLAM2::System SLAM(s, stereo ? ORB_SLAM2::System::STEREO : (mono ? ORB_SLAM2::System::MONOCULAR : ORB_SLAM2::System::RGBD), ch, depthf32);
        if (!mono) { try { sdfe::Image e; SLAM.TrackMonocular(e, 0.0); return 4; } catch (const std::runtime_error&) {} }       // wrong sensor: refused, as System.cc:329-333
        std::vector<uint8_t> a((size_t)w * h * ch), b(stereo ? (size_t)w * h * ch : (mono ? 0 : (size_t)w * h * depthElem)), m((size_t)w * h);
        for (int f = 0; f < nf; f++) {
            double ts; int32_t nb;
            if (ext) {
                int32_t st, hv; float v[16];
                if (fread(&st, 4, 1, in) != 1 || fread(&hv, 4, 1, in) != 1 || fread(v, 4, 16, in) != 16) return 5;
                SLAM.GetTracker()->mState = st;
                if (hv) { sdfe::Pose V; std::memcpy(V.m, v, 64); SLAM.GetTracker()->mVelocity = V; } else SLAM.GetTracker()->mVelocity = sdfe::Pose::none();
            }
            if (fread(&ts, 8, 1, in) != 1 || fread(&nb, 4, 1, in) != 1) return 5;
            std::vector<sdfe::Rect2d> boxes(nb > 0 ? nb : 0);
            for (int j = 0; j < nb; j++) { double r[4]; if (fread(r, 8, 4, in) != 4) return 5; boxes[j].x = r[0]; boxes[j].y = r[1]; boxes[j].width = r[2]; boxes[j].height = r[3]; }
            if (fread(a.data(), 1, a.size(), in) != a.size() || fread(b.data(), 1, b.size(), in) != b.size()) return 5;
            if (!stereo && !mono && fread(m.data(), 1, m.size(), in) != m.size()) return 5;
            sdfe::Image A, B, M;
            A.data = a.data(); A.cols = w; A.rows = h; A.nch = ch; A.step = (size_t)w * ch;
            B = A; B.data = b.data();
            if (!stereo) { B.nch = 1; B.elem = (int)depthElem; B.step = (size_t)w * depthElem; M.data = m.data(); M.cols = w; M.rows = h; M.step = (size_t)w; }
            if (mono) SLAM.TrackMonocular(A, ts);
            else if (stereo) { if (nb >= 0) SLAM.TrackStereo(A, B, boxes, ts); else SLAM.TrackStereo(A, B, ts); }
            else { if (nb >= 0) SLAM.TrackRGBD(A, B, M, boxes, ts); else SLAM.TrackRGBD(A, B, ts); }
            const ORB_SLAM2::Frame& F = SLAM.GetTracker()->mCurrentFrame;
            if (nb >= 0 && boxes.size() != F.objects.size()) return 6;          // boxTrack / firstSeparate rewrote the caller's vector
            dump(out, F);
            if (f > 0 && SLAM.GetTracker()->mLastFrame.mnId + 1 != F.mnId) return 7;
            if (ext) {
                int32_t nmp;
                if (fread(&nmp, 4, 1, in) != 1) return 5;
                if (nmp >= 0) {
                    std::vector<float> xw((size_t)nmp * 3); std::vector<uint8_t> fl((size_t)nmp);
                    if (fread(xw.data(), 4, xw.size(), in) != xw.size() || fread(fl.data(), 1, fl.size(), in) != fl.size()) return 5;
                    SLAM.GetTracker()->CommitMapPoints(xw.data(), fl.data(), nmp);
                }
            }
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    fclose(in); fclose(out);
    return 0;
}
