// Header-only mirror of the reference's C++ class API for the hot path, over the C ABI
// (include/sd_frontend.h).  Same names, argument meaning and error behaviour as
//   ORB_SLAM2::ORBextractor   include/ORBextractor.h:45-114, src/ORBextractor.cc:1043-1105
//   ORB_SLAM2::ORBmatcher::DescriptorDistance   include/ORBmatcher.h:44, src/ORBmatcher.cc:1804-1820
// OpenCV is not available in this image, so images are passed as sdfe::ImageView (data, cols, rows,
// step) and keypoints as sd_keypoint, which has cv::KeyPoint's exact 28-byte layout.  With OpenCV
// present, define SD_HAVE_OPENCV before including: the cv::InputArray / cv::OutputArray overloads
// below then make `ORB_SLAM2::ORBextractor` a drop-in for the reference class.
//
// One extractor object owns one single-image device batch, so — exactly like the reference object,
// whose operator() overwrites mvImagePyramid — it is not re-entrant: use one instance per eye
// (Tracking.cc:122-128).  For throughput use the batched C ABI directly (sd_batch_*).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "sd_frontend.h"

#ifdef SD_HAVE_OPENCV
#include <opencv2/core/core.hpp>
#endif

namespace sdfe {
struct ImageView {
    const uint8_t* data = nullptr;
    int cols = 0, rows = 0;
    size_t step = 0;
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
};
inline void check(int rc, const char* what)
{
    if (rc != SD_OK) throw std::runtime_error(std::string(what) + ": " + sd_status_string(rc) + ": " + sd_last_error());
}
}  // namespace sdfe

namespace ORB_SLAM2 {

class ORBextractor {
public:
    enum { HARRIS_SCORE = 0, FAST_SCORE = 1 };

    // One level of the public member mvImagePyramid (ORBextractor.h:85): host copy of the padded plane;
    // at(y, x) addresses the interior like cv::Mat::at on the reference's ROI (negative offsets reach the
    // 19-px BORDER_REFLECT_101 frame, as Frame::ComputeStereoMatches relies on, Frame.cc:971,988).
    struct PyramidLevel {
        std::vector<uint8_t> padded;
        int cols = 0, rows = 0;
        size_t step = 0;
        const uint8_t& at(int y, int x) const { return padded[(size_t)(y + 19) * step + (x + 19)]; }
    };

    ORBextractor(int nfeatures, float scaleFactor, int nlevels, int iniThFAST, int minThFAST)
        : nlevels_(nlevels)
    {
        sdfe::check(sd_extractor_create(&ex_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST), "ORBextractor");
        mvScaleFactor.resize(nlevels); mvInvScaleFactor.resize(nlevels);
        mvLevelSigma2.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        sd_extractor_tables(ex_, mvScaleFactor.data(), mvInvScaleFactor.data(), mvLevelSigma2.data(),
                            mvInvLevelSigma2.data(), nullptr, nullptr);
        sd_extractor_levels(ex_, nullptr, &scaleFactor_);
        mvImagePyramid.resize(nlevels);
    }
    ~ORBextractor()
    {
        if (batch_) sd_batch_destroy(batch_);
        sd_extractor_destroy(ex_);
    }
    ORBextractor(const ORBextractor&) = delete;
    ORBextractor& operator=(const ORBextractor&) = delete;

    // Compute the ORB features and descriptors on an image; mask is ignored (ORBextractor.h:56-61).
    // Empty image: returns silently leaving the outputs untouched (ORBextractor.cc:1046-1047).
    // No keypoints: descriptors is released (ORBextractor.cc:1064-1065).
    void operator()(const sdfe::ImageView& image, const sdfe::ImageView& /*mask*/, std::vector<sd_keypoint>& keypoints,
                    std::vector<uint8_t>& descriptors)
    {
        if (image.empty()) return;
        ensure_batch(image.cols, image.rows);
        sdfe::check(sd_batch_extract_host(batch_, image.data, image.step, 0, 1), "ORBextractor::operator()");
        int cap = 0, n = 0;
        sd_batch_kp_capacity(batch_, &cap);
        keypoints.resize(cap);
        descriptors.resize((size_t)cap * 32);
        sdfe::check(sd_batch_download(batch_, 0, keypoints.data(), descriptors.data(), cap, &n, nullptr), "download");
        keypoints.resize(n);
        descriptors.resize((size_t)n * 32);
        pyramidFresh_ = false;
    }

    int GetLevels() { return nlevels_; }
    float GetScaleFactor() { return scaleFactor_; }
    std::vector<float> GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    // The reference exposes the pyramid as a public data member that is always current.  Here it lives in
    // HBM; SyncPyramid() downloads it on demand (the stereo matcher reads it on the device and never needs this).
    void SyncPyramid()
    {
        if (pyramidFresh_ || !batch_) return;
        for (int l = 0; l < nlevels_; l++) {
            int w = 0, h = 0;
            sd_batch_pyramid_level(batch_, 0, l, nullptr, &w, &h, nullptr);
            PyramidLevel& P = mvImagePyramid[l];
            P.cols = w; P.rows = h; P.step = (size_t)w + 38;
            P.padded.resize(P.step * (h + 38));
            sdfe::check(sd_batch_download_pyramid(batch_, 0, l, P.padded.data()), "SyncPyramid");
        }
        pyramidFresh_ = true;
    }
    std::vector<PyramidLevel> mvImagePyramid;

    sd_extractor* handle() { return ex_; }
    sd_batch* batch() { return batch_; }

    // The reference signature's body for ANY types with OpenCV's member names: MatT has .data .cols .rows .step .empty() .type(),
    // KeyPointT is a 28-byte record laid out like cv::KeyPoint, OutArrT has create(rows, cols, type) / release() / getMat().data
    // like cv::_OutputArray.  cv::Mat, cv::KeyPoint and cv::OutputArray satisfy this; tests/cpp/cv_like.h holds stand-in types with
    // the same members so that THIS code is compiled and run on the GPU without OpenCV (tests/test_gpu_host_mirror.py).
    template <class MatT, class KeyPointT, class OutArrT>
    void extract(const MatT& image, std::vector<KeyPointT>& _keypoints, OutArrT& _descriptors)
    {
        static_assert(sizeof(KeyPointT) == sizeof(sd_keypoint), "the key-point type must have cv::KeyPoint's 28-byte layout");
        if (image.empty()) return;                                     // ORBextractor.cc:1046-1047
        if (image.type() != 0) throw std::invalid_argument("ORBextractor: image must be CV_8UC1");     // assert(image.type() == CV_8UC1), :1050
        std::vector<sd_keypoint> kps; std::vector<uint8_t> desc;
        sdfe::ImageView v; v.data = image.data; v.cols = image.cols; v.rows = image.rows; v.step = (size_t)image.step;
        (*this)(v, sdfe::ImageView(), kps, desc);
        _keypoints.resize(kps.size());
        if (!kps.empty()) std::memcpy((void*)_keypoints.data(), kps.data(), kps.size() * sizeof(sd_keypoint));
        if (kps.empty()) { _descriptors.release(); return; }           // :1064-1065
        _descriptors.create((int)kps.size(), 32, 0 /* CV_8U */);
        std::memcpy(_descriptors.getMat().data, desc.data(), desc.size());
    }

#ifdef SD_HAVE_OPENCV
    // The reference signature itself (ORBextractor.h:59): with this overload the class drops into Frame::ExtractORB (Frame.cc:655-661)
    // unchanged; it only names the OpenCV types, the body is extract() above.
    void operator()(cv::InputArray _image, cv::InputArray, std::vector<cv::KeyPoint>& _keypoints, cv::OutputArray _descriptors)
    {
        if (_image.empty()) return;
        cv::Mat image = _image.getMat();
        extract(image, _keypoints, _descriptors);
    }
#endif

protected:
    void ensure_batch(int w, int h)
    {
        if (batch_ && w == w_ && h == h_) return;
        if (batch_) { sd_batch_destroy(batch_); batch_ = nullptr; }
        sdfe::check(sd_batch_create(&batch_, ex_, w, h, 1), "sd_batch_create");
        w_ = w; h_ = h;
    }
    sd_extractor* ex_ = nullptr;
    sd_batch* batch_ = nullptr;
    int w_ = 0, h_ = 0, nlevels_ = 0;
    float scaleFactor_ = 0.f;
    bool pyramidFresh_ = false;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
};

class ORBmatcher {
public:
    static const int TH_LOW = 50, TH_HIGH = 100, HISTO_LENGTH = 30;   // ORBmatcher.cc:37-39
    // Computes the Hamming distance between two ORB descriptors (ORBmatcher.h:44)
    static int DescriptorDistance(const uint8_t* a, const uint8_t* b) { return sd_descriptor_distance(a, b); }
    // int DescriptorDistance(const cv::Mat &a, const cv::Mat &b) (ORBmatcher.h:44) for any row type with a .data pointer (cv::Mat is one)
    template <class MatT, class = typename std::enable_if<std::is_class<MatT>::value>::type>
    static int DescriptorDistance(const MatT& a, const MatT& b) { return sd_descriptor_distance((const uint8_t*)a.data, (const uint8_t*)b.data); }
};

}  // namespace ORB_SLAM2
