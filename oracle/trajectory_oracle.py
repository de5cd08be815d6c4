"""TEST INFRASTRUCTURE -- numpy restatement of the reference's trajectory writers for caller-supplied frame poses:
System::SaveTrajectoryTUM (src/System.cc:434-490) and System::SaveTrajectoryKITTI (:524-565), from `cv::Mat Tcw` on:
Rwc = Tcw[:3,:3].t(), twc = -Rwc * tcw, Converter::toQuaternion (src/Converter.cc:137-149, Eigen::Quaterniond(Matrix3d)) and the
`f << fixed << setprecision(...)` text.  PARITY UNPINNED: the reference holds no trajectory files; OpenCV's CV_32F product (double
accumulation) and Eigen's matrix -> quaternion branches are restated from the libraries' published algorithms."""
import numpy as np


def inverse_rt(Tcw):
    T = np.asarray(Tcw, np.float32).reshape(4, 4)
    Rwc = T[:3, :3].T.copy()
    twc = np.zeros(3, np.float32)
    for i in range(3):
        s = 0.0
        for k in range(3):
            s += float(-Rwc[i, k]) * float(T[k, 3])
        twc[i] = np.float32(s)
    return Rwc, twc


def to_quaternion(R):
    m = np.asarray(R, np.float64)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0.0:
        t = np.sqrt(t + 1.0)
        w = 0.5 * t
        t = 0.5 / t
        x, y, z = (m[2, 1] - m[1, 2]) * t, (m[0, 2] - m[2, 0]) * t, (m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j = (i + 1) % 3
        k = (j + 1) % 3
        t = np.sqrt(m[i, i] - m[j, j] - m[k, k] + 1.0)
        v = [0.0, 0.0, 0.0]
        v[i] = 0.5 * t
        t = 0.5 / t
        w = (m[k, j] - m[j, k]) * t
        v[j] = (m[j, i] + m[i, j]) * t
        v[k] = (m[k, i] + m[i, k]) * t
        x, y, z = v
    return [np.float32(x), np.float32(y), np.float32(z), np.float32(w)]


def tum_text(poses, stamps, lost):
    out = []
    for T, ts, l in zip(poses, stamps, lost):
        if l:
            continue
        R, t = inverse_rt(T)
        q = to_quaternion(R)
        out.append("%.6f %.9f %.9f %.9f %.9f %.9f %.9f %.9f\n" % ((ts,) + tuple(float(v) for v in t) + tuple(float(v) for v in q)))
    return "".join(out)


def kitti_text(poses):
    out = []
    for T in poses:
        R, t = inverse_rt(T)
        v = [R[0, 0], R[0, 1], R[0, 2], t[0], R[1, 0], R[1, 1], R[1, 2], t[1], R[2, 0], R[2, 1], R[2, 2], t[2]]
        out.append(" ".join("%.9f" % float(x) for x in v) + "\n")
    return "".join(out)
