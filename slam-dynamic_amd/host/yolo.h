// Header-only mirror of the reference's detector class over the C ABI (include/sd_frontend.h):
//   yolov3::yolov3Segment   include/yolo.h:22-48, src/yolo.cc:15-31 (constructor), :34-58 Segmentation, :60-77 Segmentation_
// Same members, thresholds (confThreshold 0.5, nmsThreshold 0.4, 640x480 network input) and results.  The reference's
// constructor hard-codes absolute paths under /home/hai/...; the mirror keeps that default constructor (it fails the way
// cv::dnn::readNetFromDarknet fails when the files are absent: it throws) and adds one taking the two paths.
// OpenCV is not available in this image: images are sdfe::ImageView, boxes are yolov3::Rect2d (cv::Rect2d's x, y, width,
// height doubles).  With SD_HAVE_OPENCV defined the cv::Mat overloads make the class a drop-in.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "ORBextractor.h"      // sdfe::ImageView, sdfe::check
#include "sd_frontend.h"

namespace yolov3 {

struct Rect2d { double x, y, width, height; };

// Darknet .cfg -> the layer list of the C ABI.  Only the five section types of src/yolo/yolov3.cfg are accepted.
inline void parse_darknet_cfg(const std::string& path, std::vector<sd_yolo_layer>& layers, float anchors[18], int& classes)
{
    std::ifstream f(path.c_str());
    if (!f.is_open()) throw std::runtime_error("Failed to open NetParameter file: " + path);        // readNetFromDarknet's failure
    std::string line, type;
    sd_yolo_layer cur = {};
    bool have = false, isNet = false;
    classes = 80;
    auto flush = [&]() { if (have && !isNet) layers.push_back(cur); };
    auto ints = [](const std::string& v, int32_t* out, int cap) {
        std::stringstream ss(v); std::string tok; int n = 0;
        while (std::getline(ss, tok, ',') && n < cap) out[n++] = std::stoi(tok);
        return n;
    };
    while (std::getline(f, line)) {
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line = line.substr(0, hash);
        const size_t a = line.find_first_not_of(" \t\r"), b = line.find_last_not_of(" \t\r");
        if (a == std::string::npos) continue;
        line = line.substr(a, b - a + 1);
        if (line[0] == '[') {
            flush();
            type = line.substr(1, line.size() - 2);
            cur = sd_yolo_layer(); have = true; isNet = type == "net";
            if (type == "convolutional") cur.type = SD_YOLO_CONV;
            else if (type == "shortcut") cur.type = SD_YOLO_SHORTCUT;
            else if (type == "route") cur.type = SD_YOLO_ROUTE;
            else if (type == "upsample") cur.type = SD_YOLO_UPSAMPLE;
            else if (type == "yolo") cur.type = SD_YOLO_YOLO;
            else if (!isNet) throw std::runtime_error("Unsupported Darknet layer type: " + type);
            continue;
        }
        const size_t eq = line.find('=');
        if (eq == std::string::npos || !have) continue;
        std::string k = line.substr(0, eq), v = line.substr(eq + 1);
        k.erase(k.find_last_not_of(" \t") + 1); v.erase(0, v.find_first_not_of(" \t"));
        if (isNet) continue;
        if (k == "filters") cur.filters = std::stoi(v);
        else if (k == "size") cur.size = std::stoi(v);
        else if (k == "stride") cur.stride = std::stoi(v);
        else if (k == "batch_normalize") cur.batch_normalize = std::stoi(v);
        else if (k == "activation") cur.leaky = v == "leaky";
        else if (k == "from") { cur.from[0] = std::stoi(v); cur.nfrom = 1; }
        else if (k == "layers") cur.nfrom = ints(v, cur.from, 2);
        else if (k == "mask") ints(v, cur.mask, 3);
        else if (k == "classes") classes = std::stoi(v);
        else if (k == "anchors") { std::stringstream ss(v); std::string tok; int n = 0; while (std::getline(ss, tok, ',') && n < 18) anchors[n++] = std::stof(tok); }
    }
    flush();
}

// yolov3.weights: int32 major, minor, revision; `seen` (8 bytes when major*10 + minor >= 2, else 4); then the float payload.
inline std::vector<float> read_darknet_weights(const std::string& path)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed to open NetParameter file: " + path);
    int32_t hdr[3];
    if (std::fread(hdr, 4, 3, f) != 3) { std::fclose(f); throw std::runtime_error("truncated Darknet weights header"); }
    const size_t seenBytes = (hdr[0] * 10 + hdr[1]) >= 2 ? 8 : 4;
    std::fseek(f, 0, SEEK_END);
    const long total = std::ftell(f);
    const long start = 12 + (long)seenBytes;
    std::fseek(f, start, SEEK_SET);
    std::vector<float> w((size_t)(total - start) / 4);
    if (!w.empty() && std::fread(w.data(), 4, w.size(), f) != w.size()) { std::fclose(f); throw std::runtime_error("truncated Darknet weights"); }
    std::fclose(f);
    return w;
}

class yolov3Segment {
private:
    float confThreshold = 0.5f;   // Confidence threshold
    float nmsThreshold = 0.4f;    // Non-maximum suppression threshold
    int inpWidth = 640;           // Width of network's input image
    int inpHeight = 480;          // Height of network's input image
    sd_yolo* net = nullptr;

    // precision: SD_YOLO_F32 (default) computes like cv::dnn on DNN_TARGET_CPU does (yolo.cc:29), in f32; SD_YOLO_F32W the same with the
    // 3 x 3 stride-1 layers as Winograd F(2x2, 3x3) (1.4 x faster, same tolerance and box-set tests); SD_YOLO_F32X3 the f32 operands as three bf16
    // limbs on the bf16 MFMA (1.45 x faster, same tests); SD_YOLO_F16 is the throughput mode
    // (f16 operands: 4.5 x faster, boxes can move by a pixel).
    void load(const std::string& modelConfiguration, const std::string& modelWeights, int precision)
    {
        std::vector<sd_yolo_layer> layers;
        float anchors[18] = {0};
        int classes = 80;
        parse_darknet_cfg(modelConfiguration, layers, anchors, classes);
        const std::vector<float> w = read_darknet_weights(modelWeights);
        sdfe::check(sd_yolo_create_prec(&net, layers.data(), (int)layers.size(), anchors, classes, inpWidth, inpHeight, 1, precision), "readNetFromDarknet");
        sdfe::check(sd_yolo_load_darknet_weights(net, w.data(), w.size()), "readNetFromDarknet");
    }

public:
    yolov3Segment() { load("/home/hai/projects/slam-dynamic/src/yolo/yolov3.cfg", "/home/hai/projects/slam-dynamic/src/yolo/yolov3.weights", SD_YOLO_F32); }   // yolo.cc:22-27
    yolov3Segment(const std::string& modelConfiguration, const std::string& modelWeights, int precision = SD_YOLO_F32) { load(modelConfiguration, modelWeights, precision); }
    ~yolov3Segment() { if (net) sd_yolo_destroy(net); }
    yolov3Segment(const yolov3Segment&) = delete;
    yolov3Segment& operator=(const yolov3Segment&) = delete;

    bool noTarget = true;         // whether there are masks (set by Segmentation)

    // vector<cv::Rect2d> Segmentation_(cv::Mat& image): the scaled boxes of the kept detections (yolo.cc:60-77,151-206)
    std::vector<Rect2d> Segmentation_(const sdfe::ImageView& image)
    {
        sdfe::check(sd_yolo_forward_host(net, image.data, image.cols, image.rows, image.step, confThreshold), "Segmentation_");
        double boxes[SD_MAX_BOXES * 4];
        int n = 0;
        sdfe::check(sd_yolo_boxes(net, 0, image.cols, image.rows, confThreshold, nmsThreshold, boxes, nullptr, nullptr, SD_MAX_BOXES, &n), "Segmentation_");
        std::vector<Rect2d> out((size_t)n);
        for (int i = 0; i < n; i++) out[i] = Rect2d{boxes[4 * i], boxes[4 * i + 1], boxes[4 * i + 2], boxes[4 * i + 3]};
        return out;
    }

    // cv::Mat Segmentation(cv::Mat& image): rows x cols u8, 1 = keep, 0 = inside the dilated box cores (yolo.cc:34-58)
    std::vector<uint8_t> Segmentation(const sdfe::ImageView& image)
    {
        sdfe::check(sd_yolo_forward_host(net, image.data, image.cols, image.rows, image.step, confThreshold), "Segmentation");
        std::vector<uint8_t> mask((size_t)image.cols * image.rows);
        int nt = 1;
        sdfe::check(sd_yolo_mask_host(net, image.cols, image.rows, confThreshold, nmsThreshold, mask.data(), (size_t)image.cols, &nt), "Segmentation");
        noTarget = nt != 0;
        return mask;
    }

    // vector<cv::Rect2d> Segmentation_(cv::Mat&) / cv::Mat Segmentation(cv::Mat&) (include/yolo.h:34-35) for ANY types with OpenCV's member
    // names (RectT(x, y, width, height); MatT with .data .cols .rows .step and a (rows, cols, type) constructor): cv::Rect2d / cv::Mat in a
    // build that has OpenCV, tests/cpp/cv_like.h's stand-ins here, where this code is compiled and run by tests/test_gpu_host_mirror.py.
    template <class RectT, class MatT> std::vector<RectT> SegmentationRects(MatT& image)
    {
        sdfe::ImageView v; v.data = image.data; v.cols = image.cols; v.rows = image.rows; v.step = (size_t)image.step;
        std::vector<RectT> out;
        for (const Rect2d& r : Segmentation_(v)) out.push_back(RectT(r.x, r.y, r.width, r.height));
        return out;
    }
    template <class MatT> MatT SegmentationMat(MatT& image)
    {
        sdfe::ImageView v; v.data = image.data; v.cols = image.cols; v.rows = image.rows; v.step = (size_t)image.step;
        const std::vector<uint8_t> m = Segmentation(v);
        MatT out(image.rows, image.cols, 0 /* CV_8U */);
        for (int y = 0; y < image.rows; y++) std::memcpy(out.data + (size_t)y * (size_t)out.step, m.data() + (size_t)y * image.cols, (size_t)image.cols);
        return out;
    }
#ifdef SD_HAVE_OPENCV
    std::vector<cv::Rect2d> Segmentation_(cv::Mat& image) { return SegmentationRects<cv::Rect2d>(image); }
    cv::Mat Segmentation(cv::Mat& image) { return SegmentationMat(image); }
#endif
};

}  // namespace yolov3
