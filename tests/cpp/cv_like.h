// Test doubles with the member names of the OpenCV types the host mirrors are written against (cv::Mat, cv::KeyPoint, cv::Rect2d,
// cv::_OutputArray).  OpenCV does not exist in this image; these let the mirrors' template bodies -- the code an OpenCV build runs --
// be compiled and executed here.  They are test scaffolding, not a substitute for OpenCV and not used to build anything of the reference.
#pragma once
#include <cstdint>
#include <vector>
namespace cvlike {
struct Point2f { float x, y; };
struct KeyPoint { Point2f pt; float size, angle, response; int octave, class_id; };      // cv::KeyPoint's layout (28 bytes)
struct Rect2d { double x, y, width, height; Rect2d(double x_, double y_, double w_, double h_) : x(x_), y(y_), width(w_), height(h_) {} };
struct Mat {
    uint8_t* data = nullptr; int cols = 0, rows = 0; size_t step = 0; int type_ = 0;
    std::vector<uint8_t> own;
    Mat() {}
    Mat(int r, int c, int t) : cols(c), rows(r), step((size_t)c), type_(t), own((size_t)r * c) { data = own.data(); }
    Mat(int r, int c, int t, uint8_t* p, size_t s) : data(p), cols(c), rows(r), step(s), type_(t) {}
    bool empty() const { return !data || cols <= 0 || rows <= 0; }
    int type() const { return type_; }
    int channels() const { return 1; }
};
struct OutputArray {                                                                       // cv::_OutputArray's three members used by the mirror
    mutable Mat m;
    void create(int rows, int cols, int type) const { m = Mat(rows, cols, type); m.data = m.own.data(); }
    void release() const { m = Mat(); }
    Mat& getMat() const { return m; }
};
}  // namespace cvlike
