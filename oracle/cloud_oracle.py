"""TEST INFRASTRUCTURE -- numpy restatement of PointCloudMapping::generatePointCloud (src/pointcloudmapping.cc:59-103) and of
the dyn_boxes filter of Tracking::CreateNewKeyFrame (src/Tracking.cc:1999-2007).  PCL is a third-party dependency that is not
under /root/reference: pcl::transformPointCloud is restated as the f64 affine map, evaluated left to right and narrowed to f32
[PCL-recall]; the reference holds no fixture for this path -> parity unpinned."""
import numpy as np

CLOUD_POINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("b", "u1"), ("g", "u1"), ("r", "u1"), ("a", "u1")])


def dyn_boxes(objects, box_status):
    """Tracking.cc:2000-2005: objects whose box_status is 2 or 0."""
    st = np.asarray(box_status)
    return np.asarray(objects, np.float64).reshape(-1, 4)[(st == 2) | (st == 0)]


def generate_point_cloud(color, depth_u16, depth_factor, mask_u8, boxes, fx, fy, cx, cy, Twc):
    """-> (points CLOUD_POINT_DTYPE in push_back order, masked_num).  color (H, W, 3) u8, depth (H, W) u16 scaled by depth_factor
    (imDepth.convertTo(CV_32F, factor)), mask (H, W) u8 or None, boxes (k, 4) f64 x, y, w, h, Twc 4x4 f64 = T.inverse()."""
    H, W = depth_u16.shape
    ms, ns = np.arange(0, H, 3), np.arange(0, W, 3)
    mm, nn = np.meshgrid(ms, ns, indexing="ij")
    d = depth_u16[mm, nn].astype(np.float32) * np.float32(depth_factor)
    inbox = np.zeros(mm.shape, bool)
    px, py = nn.astype(np.float32).astype(np.float64), mm.astype(np.float32).astype(np.float64)     # cv::Point2f(n, m) -> Rect2d::contains
    for (x, y, w, h) in np.asarray(boxes, np.float64).reshape(-1, 4):
        inbox |= (x <= px) & (px < x + w) & (y <= py) & (py < y + h)
    skip = inbox & (mask_u8[mm, nn] != 0) if mask_u8 is not None else np.zeros(mm.shape, bool)
    masked_num = int(skip.sum())
    keep = ~((d.astype(np.float64) < 0.01) | (d > np.float32(5)) | skip)
    z = d[keep]
    n = nn[keep].astype(np.float32); m = mm[keep].astype(np.float32)
    x = ((n - np.float32(cx)) * z) / np.float32(fx)
    y = ((m - np.float32(cy)) * z) / np.float32(fy)
    T = np.asarray(Twc, np.float64).reshape(4, 4)
    xd, yd, zd = x.astype(np.float64), y.astype(np.float64), z.astype(np.float64)
    out = np.zeros(len(z), CLOUD_POINT_DTYPE)
    for k, name in enumerate(("x", "y", "z")):
        out[name] = (((T[k, 0] * xd + T[k, 1] * yd) + T[k, 2] * zd) + T[k, 3]).astype(np.float32)
    c = color[mm[keep], nn[keep]]
    out["b"], out["g"], out["r"], out["a"] = c[:, 0], c[:, 1], c[:, 2], 255
    return out, masked_num
