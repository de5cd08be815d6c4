// Header-only mirror of the reference's trajectory writers (src/System.cc:434-565) for a caller that supplies the frame poses itself:
//   ORB_SLAM2::System::SaveTrajectoryTUM    :434-490   "timestamp tx ty tz qx qy qz qw", lost frames skipped
//   ORB_SLAM2::System::SaveTrajectoryKITTI  :524-565   the 3 x 4 matrix [Rwc | twc] row by row, every frame
// In the reference a frame's pose is rebuilt from its reference key frame (mlRelativeFramePoses * key-frame pose * Two), i.e. from
// back-end state that is outside this front end; what is mirrored here is everything AFTER that product -- Tcw -> (Rwc, twc) ->
// quaternion -> text -- so that the dense mapper's / tracker's output can be evaluated with `evo` the way README.md:67-77 does.
// No GPU and no library call: plain C++ (tests/test_trajectory_writers.py compiles and checks it on the CPU).
#pragma once
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iomanip>
#include <string>
#include <vector>

namespace sdfe {

struct TrajectoryPose { float Tcw[16]; double timestamp; bool lost; };       // row-major 4x4 CV_32F, mlFrameTimes, mlbLost

// Tcw.rowRange(0,3).colRange(0,3).t() and -Rwc*tcw [OpenCV-recall: the CV_32F product accumulates in double, k ascending, narrowed once]
inline void inverse_rt(const float* Tcw, float Rwc[9], float twc[3])
{
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rwc[3 * i + j] = Tcw[4 * j + i];
    for (int i = 0; i < 3; i++) {
        double s = 0.0;
        for (int k = 0; k < 3; k++) s += (double)(-Rwc[3 * i + k]) * (double)Tcw[4 * k + 3];
        twc[i] = (float)s;
    }
}

// Converter::toQuaternion (src/Converter.cc:137-149): Eigen::Quaterniond(Matrix3d) [Eigen-recall: trace branch, else the largest diagonal
// element], returned as floats x, y, z, w.
inline void to_quaternion(const float R[9], float q[4])
{
    double m[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) m[i][j] = (double)R[3 * i + j];
    double x, y, z, w;
    double t = m[0][0] + m[1][1] + m[2][2];
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        w = 0.5 * t;
        t = 0.5 / t;
        x = (m[2][1] - m[1][2]) * t; y = (m[0][2] - m[2][0]) * t; z = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        double v[3];
        v[i] = 0.5 * t;
        t = 0.5 / t;
        w = (m[k][j] - m[j][k]) * t;
        v[j] = (m[j][i] + m[i][j]) * t;
        v[k] = (m[k][i] + m[i][k]) * t;
        x = v[0]; y = v[1]; z = v[2];
    }
    q[0] = (float)x; q[1] = (float)y; q[2] = (float)z; q[3] = (float)w;
}

// System::SaveTrajectoryTUM (src/System.cc:434-490)
inline bool SaveTrajectoryTUM(const std::string& filename, const std::vector<TrajectoryPose>& poses)
{
    std::ofstream f(filename.c_str());
    if (!f.is_open()) return false;
    f << std::fixed;
    for (const TrajectoryPose& p : poses) {
        if (p.lost) continue;
        float Rwc[9], twc[3], q[4];
        inverse_rt(p.Tcw, Rwc, twc);
        to_quaternion(Rwc, q);
        f << std::setprecision(6) << p.timestamp << " " << std::setprecision(9) << twc[0] << " " << twc[1] << " " << twc[2] << " " << q[0] << " " << q[1] << " "
          << q[2] << " " << q[3] << std::endl;
    }
    return true;
}

// System::SaveTrajectoryKITTI (src/System.cc:524-565): every frame, lost or not
inline bool SaveTrajectoryKITTI(const std::string& filename, const std::vector<TrajectoryPose>& poses)
{
    std::ofstream f(filename.c_str());
    if (!f.is_open()) return false;
    f << std::fixed;
    for (const TrajectoryPose& p : poses) {
        float Rwc[9], twc[3];
        inverse_rt(p.Tcw, Rwc, twc);
        f << std::setprecision(9) << Rwc[0] << " " << Rwc[1] << " " << Rwc[2] << " " << twc[0] << " " << Rwc[3] << " " << Rwc[4] << " " << Rwc[5] << " " << twc[1] << " "
          << Rwc[6] << " " << Rwc[7] << " " << Rwc[8] << " " << twc[2] << std::endl;
    }
    return true;
}

}  // namespace sdfe
