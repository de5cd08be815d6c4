// Exercises slam-dynamic_amd/host/yolo.h (the mirror of yolov3::yolov3Segment):
//   yolo_mirror_main <cfg> <weights> <w> <h> <bgr.raw> <out.bin> [precision: absent = the class default (f32), else SD_YOLO_*]
// writes: int32 n, n x 4 doubles (Segmentation_), int32 noTarget, w*h mask bytes (Segmentation).
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>
#include "yolo.h"
#include "cv_like.h"
#include <cstring>

int main(int argc, char** argv)
{
    if (argc < 7) return 2;
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    std::vector<uint8_t> img((size_t)w * h * 3);
    FILE* f = fopen(argv[5], "rb");
    if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
    fclose(f);
    bool threw = false;
    try { yolov3::yolov3Segment missing; } catch (const std::exception&) { threw = true; }      // the reference's hard-coded paths do not exist here
    if (!threw) return 4;
    try {
        std::unique_ptr<yolov3::yolov3Segment> py(argc > 7 ? new yolov3::yolov3Segment(argv[1], argv[2], atoi(argv[7])) : new yolov3::yolov3Segment(argv[1], argv[2]));
        yolov3::yolov3Segment& yolo = *py;
        sdfe::ImageView v; v.data = img.data(); v.cols = w; v.rows = h; v.step = (size_t)w * 3;
        const std::vector<yolov3::Rect2d> boxes = yolo.Segmentation_(v);
        const std::vector<uint8_t> mask = yolo.Segmentation(v);
        {   // the cv-typed entry points' bodies (what an OpenCV build runs), with stand-in types of the same member names
            cvlike::Mat cimg(h, w, 16 /* CV_8UC3 */, img.data(), (size_t)w * 3);
            const std::vector<cvlike::Rect2d> cb = yolo.SegmentationRects<cvlike::Rect2d>(cimg);
            if (cb.size() != boxes.size()) return 5;
            for (size_t i = 0; i < cb.size(); i++)
                if (cb[i].x != boxes[i].x || cb[i].y != boxes[i].y || cb[i].width != boxes[i].width || cb[i].height != boxes[i].height) return 6;
            const cvlike::Mat cm = yolo.SegmentationMat(cimg);
            if (cm.rows != h || cm.cols != w || memcmp(cm.data, mask.data(), mask.size()) != 0) return 7;
        }
        FILE* o = fopen(argv[6], "wb");
        const int32_t n = (int32_t)boxes.size();
        fwrite(&n, 4, 1, o);
        for (const yolov3::Rect2d& r : boxes) fwrite(&r, sizeof(double), 4, o);
        const int32_t nt = yolo.noTarget ? 1 : 0;
        fwrite(&nt, 4, 1, o);
        fwrite(mask.data(), 1, mask.size(), o);
        fclose(o);
    } catch (const std::exception& e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
    return 0;
}
