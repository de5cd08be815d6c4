/* sd_frontend.h — C ABI of the MI355X-native per-frame front end (libsd_frontend.so).
 *
 * Drop-in boundary for the hot path of li-guihai/slam-dynamic (an ORB-SLAM2 fork):
 * ORBextractor, the Frame-side stereo/RGB-D association, the 256-bit Hamming matchers
 * and the dynamic-point cull.  The reference has no FFI layer — its boundary is the C++
 * class API in namespace ORB_SLAM2 — so every entry point below names the reference
 * interface it replaces (file:line relative to the reference tree).  Plain pointers and
 * sizes only; no OpenCV, torch or STL types.  All functions return an int status
 * (SD_OK == 0, < 0 on error) and never throw or abort.  The reference-side bindings a
 * maintainer would add are shown in INTEGRATION.md and in
 * slam-dynamic_amd/host/ORBextractor.h (a header-only mirror of the class API).
 *
 * Device model: one process per GPU.  Pointers named d_* are device (HBM) pointers;
 * `stream` is a hipStream_t passed as void* (NULL = the library's own stream).
 */
#ifndef SD_FRONTEND_H
#define SD_FRONTEND_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_OK 0
#define SD_ERR_INVALID (-1)     /* bad argument                                             */
#define SD_ERR_NO_DEVICE (-2)   /* no HIP device / HIP runtime unusable (never a CPU fallback) */
#define SD_ERR_HIP (-3)         /* a HIP call failed; see sd_last_error()                   */
#define SD_ERR_CAPACITY (-4)    /* caller buffer too small                                  */
#define SD_ERR_UNSUPPORTED (-5) /* geometry outside what the kernels were sized for         */
#define SD_ERR_STATE (-6)       /* call sequence error (e.g. stereo before extract)         */

/* Same 28-byte layout as cv::KeyPoint {Point2f pt; float size, angle, response; int octave, class_id;}
 * so an adapter can memcpy into std::vector<cv::KeyPoint> (ORBextractor.h:59-61). */
typedef struct sd_keypoint {
    float x, y, size, angle, response;
    int32_t octave, class_id;
} sd_keypoint;

typedef struct sd_extractor sd_extractor; /* ORB_SLAM2::ORBextractor state (parameters, tables)      */
typedef struct sd_batch sd_batch;         /* device workspace for up to max_images images of one size */

int sd_version(void);
const char* sd_status_string(int status);
const char* sd_last_error(void); /* thread-local text of the last SD_ERR_HIP / SD_ERR_UNSUPPORTED */
int sd_device_count(int* n);     /* hipGetDeviceCount; SD_ERR_NO_DEVICE when none */

/* ---- ORBextractor (include/ORBextractor.h:45-114, src/ORBextractor.cc:410-470) ---- */

/* ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)  ORBextractor.h:51 */
int sd_extractor_create(sd_extractor** out, int nfeatures, float scaleFactor, int nlevels, int iniThFAST,
                        int minThFAST);
int sd_extractor_destroy(sd_extractor* ex);
/* The 7 fixed-point (8.8) taps of GaussianBlur(7x7, sigma 2) (ORBextractor.cc:1086) are a spec
 * parameter (OpenCV-version dependent); default {18,34,48,56,48,34,18}. */
int sd_extractor_set_blur_taps(sd_extractor* ex, const uint16_t taps[7]);
/* GetLevels / GetScaleFactor / GetScaleFactors / GetInverseScaleFactors / GetScaleSigmaSquares /
 * GetInverseScaleSigmaSquares  (ORBextractor.h:63-83) + mnFeaturesPerLevel, umax (ORBextractor.h:102-104).
 * Any pointer may be NULL.  scale..inv_sigma2, quota: nlevels entries; umax: 16 entries. */
int sd_extractor_levels(const sd_extractor* ex, int* nlevels, float* scale_factor);
int sd_extractor_tables(const sd_extractor* ex, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                        int32_t* quota, int32_t* umax);
/* Size of mvImagePyramid[level] for a WxH input (ORBextractor.cc:1111-1112). */
int sd_extractor_level_size(const sd_extractor* ex, int width, int height, int level, int* lw, int* lh);

/* ---- batched ORBextractor::operator() (src/ORBextractor.cc:1043-1105) ----
 * One sd_batch owns, in HBM, the 8-level padded pyramid (the public member mvImagePyramid,
 * ORBextractor.h:85), the blurred planes, FAST candidates and the results of up to max_images
 * images of width x height.  Like the reference object it is not re-entrant: one batch per
 * concurrent caller (the reference uses one extractor per eye, Frame.cc:87-90). */
int sd_batch_create(sd_batch** out, sd_extractor* ex, int width, int height, int max_images);
int sd_batch_destroy(sd_batch* b);
int sd_batch_kp_capacity(const sd_batch* b, int* cap); /* max keypoints one image can return */

/* operator()(image, mask [ignored], keypoints, descriptors) for n_images gray images already in
 * HBM: image i starts at d_gray + i*image_pitch, rows `stride` bytes apart.  Asynchronous on
 * `stream`; results stay on the device. */
int sd_batch_extract_device(sd_batch* b, const uint8_t* d_gray, size_t stride, size_t image_pitch, int n_images,
                            void* stream);
/* The same for 3-channel 8-bit input (BGR, or RGB when rgb_order != 0: Camera.RGB): Tracking::GrabImageRGBD / GrabImageStereo's
 * cvtColor (src/Tracking.cc:179-198,259-268) fused into the extractor's level-0 copy (src/ORBextractor.cc:1127-1128), so the
 * gray image is never stored; it is the interior of pyramid level 0 (sd_batch_pyramid_level).  Results are identical to
 * sd_cvt_gray_device followed by sd_batch_extract_device.  `stride` in bytes (>= 3 * width). */
int sd_batch_extract_color_device(sd_batch* b, const uint8_t* d_src, size_t stride, size_t image_pitch, int rgb_order, int n_images,
                                  void* stream);
/* The same for any input Tracking::GrabImage* accepts (src/Tracking.cc:175-200, 187-200: `if(mImGray.channels()==3) ... else
 * if(mImGray.channels()==4)` with CV_RGB2GRAY / CV_BGR2GRAY / CV_RGBA2GRAY / CV_BGRA2GRAY): channels = 1 (gray, copied), 3, or 4 (the
 * alpha byte is ignored, same weights).  `stride` in bytes (>= channels * width). */
int sd_batch_extract_pixels_device(sd_batch* b, const uint8_t* d_src, size_t stride, size_t image_pitch, int channels, int rgb_order,
                                   int n_images, void* stream);
/* Same, from host memory (upload + extract + stream sync).  Empty image (NULL / 0 size) => 0 keypoints,
 * as ORBextractor.cc:1046-1047. */
int sd_batch_extract_host(sd_batch* b, const uint8_t* gray, size_t stride, size_t image_pitch, int n_images);

/* Results.  Device views: kp [max_images][cap], desc [max_images][cap][32], count [max_images]. */
int sd_batch_results_device(sd_batch* b, sd_keypoint** d_kp, uint8_t** d_desc, int32_t** d_count, int* cap);
int sd_batch_counts(sd_batch* b, int32_t* counts, int n_images);           /* syncs the stream */
int sd_batch_download(sd_batch* b, int image, sd_keypoint* kp, uint8_t* desc, int cap, int* n,
                      int32_t* per_level /* nlevels or NULL */);
/* mvImagePyramid[level] of image `image`: device pointer to the interior (x=0,y=0) pixel; the 19-px
 * BORDER_REFLECT_101 frame lies at negative offsets (ORBextractor.cc:1107-1130). */
int sd_batch_pyramid_level(sd_batch* b, int image, int level, const uint8_t** d_interior, int* w, int* h,
                           size_t* stride);
/* Copy the padded plane ((h+38) x (w+38), tightly packed) / the blurred interior plane (h x w) to host. */
int sd_batch_download_pyramid(sd_batch* b, int image, int level, uint8_t* padded_out);
int sd_batch_download_blurred(sd_batch* b, int image, int level, uint8_t* out);
/* Number of FAST candidates per level before the quadtree (vToDistributeKeys.size(), ORBextractor.cc:829) */
int sd_batch_candidate_counts(sd_batch* b, int image, int32_t* per_level);

/* ---- Frame::ComputeStereoMatches (src/Frame.cc:874-1048) ----
 * The batch must hold n_frames stereo pairs extracted as images 2f (left) and 2f+1 (right).
 * mbf = Camera.bf, fx = Camera.fx (mb = mbf/fx, Frame.cc:228).  Outputs per left keypoint, stored at the
 * LEFT image's slot (image index 2f) of the [max_images][cap] arrays. */
int sd_batch_stereo_match(sd_batch* b, int n_frames, float mbf, float fx, void* stream);
int sd_batch_stereo_device(sd_batch* b, float** d_uright, float** d_depth, int* cap); /* [max_images][cap] */
int sd_batch_download_stereo(sd_batch* b, int frame, float* uright, float* depth, int32_t* sad_dist, int cap);

/* ---- Frame::ComputeStereoFromRGBD (src/Frame.cc:1051-1072) + depth scaling (Tracking.cc:271-272) ----
 * d_depth: 16-bit depth images (rows `stride_elems` elements apart); depth_factor = 1/DepthMapFactor.
 * Fuses imDepth.convertTo(CV_32F, factor) with the per-keypoint lookup (undistortion is the identity
 * when k1 == 0, Frame.cc:814-818).  One image per frame. */
int sd_batch_rgbd_from_u16(sd_batch* b, const uint16_t* d_depth, size_t stride_elems, size_t image_pitch_elems,
                           int n_images, float depth_factor, float mbf, void* stream);
int sd_batch_rgbd_from_f32(sd_batch* b, const float* d_depth, size_t stride_elems, size_t image_pitch_elems,
                           int n_images, float mbf, void* stream);
/* CV_32F depth that still has to be scaled: `if((fabs(mDepthMapFactor-1.0f)>1e-5) || imDepth.type()!=CV_32F) imDepth.convertTo(imDepth,
 * CV_32F, mDepthMapFactor)` (Tracking.cc:271-272) for a CV_32F input and a factor other than 1: d = depth * depth_factor in f32. */
int sd_batch_rgbd_from_f32_scaled(sd_batch* b, const float* d_depth, size_t stride_elems, size_t image_pitch_elems, int n_images,
                                  float depth_factor, float mbf, void* stream);
int sd_batch_download_rgbd(sd_batch* b, int image, float* uright, float* depth, int cap);


/* ---- Frame statics (include/Frame.h:160-210): intrinsics, stereo baseline, undistorted image bounds ---- */
typedef struct sd_camera {
    float fx, fy, cx, cy, mbf, mb;               /* mb = mbf/fx (Frame.cc:228) */
    float mnMinX, mnMaxX, mnMinY, mnMaxY;        /* Frame::ComputeImageBounds (Frame.cc:844-872) */
} sd_camera;

/* Frame::AssignFeaturesToGrid / PosInGrid (src/Frame.cc:463-478,790-800): the 64x48 grid cell of every
 * keypoint (posX*48 + posY, or -1 outside).  The device form of mGrid: per-cell lists are implied by
 * (cell, keypoint index) order, which is the visiting order of GetFeaturesInArea (Frame.cc:735-788). */
int sd_batch_assign_grid(sd_batch* b, int n_images, const sd_camera* cam, void* stream);
int sd_batch_download_grid(sd_batch* b, int image, int16_t* cell, int cap);

/* Frame::UnprojectStereo (src/Frame.cc:1074-1088) for every keypoint of frames first_image + k*image_step,
 * k < n_frames, using the depth of the preceding stereo / RGB-D step.  Twc_host: n_frames row-major 4x4
 * [mRwc | mOw].  Fills the batch's map-point table: world position + flags (bit0 valid, bit1 = the map
 * point has Observations() > 0, never set here).  first_image must be 0. */
int sd_batch_unproject(sd_batch* b, int first_image, int image_step, int n_frames, const sd_camera* cam,
                       const float* Twc_host, void* stream);
/* The map-point table ([max_images][cap][3] f32, [max_images][cap] u8): a caller that owns real MapPoints
 * (Tracking) writes LastFrame.mvpMapPoints[i]->GetWorldPos() / flags here instead of calling unproject. */
int sd_batch_mappoints_device(sd_batch* b, float** d_xw, uint8_t** d_flags, int* cap);
int sd_batch_set_mappoints(sd_batch* b, int image, const float* xw, const uint8_t* flags, int n);
int sd_batch_download_mappoints(sd_batch* b, int image, float* xw, uint8_t* flags, int cap);

/* ORBmatcher::SearchByProjection(Frame& Current, const Frame& Last, th, bMono[, points_last, points_current])
 * (src/ORBmatcher.cc:1485-1627 and the pair-emitting overload :407-559, called from Tracking::TrackHomo,
 * Tracking.cc:998-1010, and TrackWithMotionModel).  Pair p matches Current = slot cur_index[p] against
 * Last = slot last_index[p] of the same batch (host index arrays; sd_batch_assign_grid must have run on the
 * Current slots).  Tcw_host / Tlw_host: n_pairs row-major 4x4 poses (CurrentFrame.mTcw, LastFrame.mTcw).
 * d_occupied (nullable, [n_pairs][cap] u8): CurrentFrame.mvpMapPoints[i2] already holds a point with
 * Observations() > 0.  d_mp_desc (nullable, [max_images][cap][32]): pMP->GetDescriptor() of the Last frame's
 * points; NULL = the Last frame's own descriptors.
 * Results: match[i2] = index of the Last-frame point assigned to Current keypoint i2 or -1 (the new
 * CurrentFrame.mvpMapPoints); pairs = (i, i2) in order of i == points_last / points_current BEFORE the
 * rotation-histogram cull (ORBmatcher.cc:505-506); nmatches = the function's return value. */
int sd_batch_search_by_projection(sd_batch* b, int n_pairs, const int32_t* cur_index, const int32_t* last_index,
                                  const float* Tcw_host, const float* Tlw_host, const sd_camera* cam, float th, int bMono,
                                  int checkOrientation, const uint8_t* d_occupied, const uint8_t* d_mp_desc, void* stream);
/* Tracking::SearchLocalPoints (src/Tracking.cc:2014-2064) = Frame::isInFrustum(pMP, viewing_cos_limit)
 * (src/Frame.cc:677-733) for every local map point, then ORBmatcher::SearchByProjection(Frame& F, const
 * vector<MapPoint*>& vpMapPoints, th) with mfNNratio = nnratio (src/ORBmatcher.cc:45-129; the caller passes th = 1,
 * 3 for RGB-D, 5 after a relocalisation, Tracking.cc:2055-2063).  Frame f < n_frames is batch slot frame_index[f]
 * (sd_batch_assign_grid must have run on it) with local map points [point_offset[f], point_offset[f+1]) of the flat
 * device arrays d_points / d_point_desc (pMP->GetDescriptor(), 32 B each), pose Tcw_host[f] (row-major 4x4 mTcw).
 * d_occupied (nullable, [n_frames][cap] u8): F.mvpMapPoints[i] already holds a point with Observations() > 0.
 * Outputs (device): d_track[m] = mbTrackInView / mTrackProjX / mTrackProjY / mTrackProjXR / mnTrackScaleLevel /
 * mTrackViewCos; d_point_match[m] = keypoint index given to point m or -1; d_kp_match[f][i] (row stride =
 * sd_batch cap) = index, relative to point_offset[f], of the point this call leaves in F.mvpMapPoints[i], or -1;
 * d_nmatches[f] = the function's return value. */
typedef struct sd_map_point {
    float xw[3];          /* GetWorldPos() */
    float normal[3];      /* GetNormal() */
    float min_distance;   /* mfMinDistance (GetMinDistanceInvariance() / 0.8f) */
    float max_distance;   /* mfMaxDistance (GetMaxDistanceInvariance() / 1.2f) */
    uint32_t flags;       /* bit0: !isBad() && mnLastFrameSeen != F.mnId; bit1: Observations() > 0 */
} sd_map_point;           /* 36 bytes */
typedef struct sd_track_info { float proj_x, proj_y, proj_xr, view_cos; int32_t level; int32_t in_view; } sd_track_info;   /* 24 bytes */
int sd_batch_search_local_map(sd_batch* b, int n_frames, const int32_t* frame_index, const int32_t* point_offset,
                              const sd_map_point* d_points, const uint8_t* d_point_desc, const float* Tcw_host,
                              const sd_camera* cam, float th, float nnratio, float viewing_cos_limit,
                              const uint8_t* d_occupied, sd_track_info* d_track, int32_t* d_point_match,
                              int32_t* d_kp_match, int32_t* d_nmatches, void* stream);
/* ---- vocabulary + bag of words (Thirdparty/DBoW2, src/Frame.cc:803-810, src/ORBmatcher.cc:159-288) ----
 * sd_vocab = ORBVocabulary (DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>): one packed buffer in HBM
 * (header | desc[n][32] | weight[n] f64 | parent[n] | childStart[n+1] | childIdx[n-1] | wordId[n]; node ids and the
 * order of children are the text file's).  The buffer is what a multi-GPU job broadcasts once at start-up:
 * rank 0 loads, every rank allocates sd_vocab_packed_bytes(n_nodes), RCCL-broadcasts into it and adopts it with
 * sd_vocab_from_packed_device (the buffer stays owned by the caller). */
typedef struct sd_vocab sd_vocab;
/* TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1338-1424; System.cc:70-78 loads ORBvoc.txt with it).
 * Errors: unreadable file or a header outside k<=20, 1<=L<=10, scoring<=5, weighting<=3 -> SD_ERR_INVALID with the
 * reference's message. */
int sd_vocab_load_text(sd_vocab** out, const char* path);
/* The same from parsed node lines (line i = node i+1): parent id, isLeaf, 32 descriptor bytes, weight. */
int sd_vocab_from_nodes(sd_vocab** out, int k, int L, int scoring, int weighting, int n_lines, const int32_t* parent,
                        const uint8_t* is_leaf, const uint8_t* desc, const double* weight);
int sd_vocab_from_packed_device(sd_vocab** out, void* d_blob, size_t bytes);
void sd_vocab_destroy(sd_vocab* v);
int sd_vocab_info(const sd_vocab* v, int* k, int* L, int* scoring, int* weighting, int* n_nodes, int* n_words);
int sd_vocab_packed_device(sd_vocab* v, void** d_blob, size_t* bytes);
size_t sd_vocab_packed_bytes(int n_nodes);
int sd_vocab_download_nodes(const sd_vocab* v, int32_t* parent, int32_t* n_children, int32_t* word_id, uint8_t* desc, double* weight);

/* Frame::ComputeBoW (src/Frame.cc:803-810): mpORBvocabulary->transform(descriptors, mBowVec, mFeatVec, levelsup = 4)
 * for the listed image slots.  Per slot: the BowVector (distinct word ids ascending + f64 values, normalised as the
 * vocabulary's scoring asks), the FeatureVector flattened in map order (node id at level L - levelsup, feature index),
 * and per feature the word / weight / node of TemplatedVocabulary::transform(feature, ...).
 * meta[image][4] = {listed features, distinct nodes, distinct words, 0}. */
int sd_batch_compute_bow(sd_batch* b, const sd_vocab* v, int n_images, const int32_t* image_index, int levelsup, void* stream);
int sd_batch_bow_device(sd_batch* b, uint32_t** d_bow_word, double** d_bow_value, uint32_t** d_fv_node, uint32_t** d_fv_feature,
                        int32_t** d_meta, int* cap);
int sd_batch_download_bow(sd_batch* b, int image, uint32_t* bow_word, double* bow_value, int* n_words, uint32_t* fv_node,
                          uint32_t* fv_feature, int* n_features, uint32_t* feature_word, double* feature_weight,
                          uint32_t* feature_node, int cap);
/* ORBmatcher::SearchByBoW(KeyFrame* pKF, Frame& F, vector<MapPoint*>& vpMapPointMatches) (src/ORBmatcher.cc:159-288;
 * callers Tracking::TrackReferenceKeyFrame and Relocalization): pair p matches keyframe slot kf_index[p] against frame
 * slot frame_index[p] (sd_batch_compute_bow must have run on both).  d_kf_valid (nullable, [n_pairs][cap] u8):
 * pKF->GetMapPointMatches()[i] != NULL && !isBad().  Results through sd_batch_download_matches: match[iF] = index of the
 * keyframe feature whose map point lands in vpMapPointMatches[iF] (or -1), nmatches = the return value; no pairs. */
int sd_batch_search_by_bow(sd_batch* b, int n_pairs, const int32_t* kf_index, const int32_t* frame_index,
                           const uint8_t* d_kf_valid, float nnratio, int checkOrientation, void* stream);

/* The model fit of Tracking::TrackHomo (src/Tracking.cc:1026-1075) for every pair of the preceding
 * sd_batch_search_by_projection, from its points_last / points_current: replaces cv::findHomography(points_last,
 * points_current, RANSAC, 3, inliers_H), cv::findFundamentalMat(..., RANSAC, 3, 0.99, inliers_F), the inlier counts
 * and the choice `n_H > n_F ? (H, return 1) : (F, return 2)` when either count exceeds 10 (else 0).  The estimator is
 * this library's own specification (OpenCV's cannot be matched bit for bit without OpenCV); HorF / flag
 * are what sd_batch_separate takes.  Masks are indexed like the pair list of sd_batch_download_matches. */
int sd_batch_estimate_motion(sd_batch* b, void* stream);
int sd_batch_download_motion(sd_batch* b, int pair, double* H, double* F, uint8_t* mask_h, uint8_t* mask_f, int cap,
                             int* n_points, int* n_h, int* n_f, float* HorF, int* flag);

/* Frame::UndistortKeyPoints (src/Frame.cc:812-842) and Frame::ComputeImageBounds (:844-872) for cameras whose
 * Camera.k1 is not zero (TUM1 / TUM2 / EuRoC settings): cv::undistortPoints(pts, mK, mDistCoef, Mat(), mK).  K4 = fx, fy,
 * cx, cy; dist5 = k1, k2, p1, p2, k3 (Tracking.cc:76-90).  sd_batch_undistort_keypoints writes the mvKeysUn records of the
 * first n_images slots to d_keys_un ([n_images][cap] sd_keypoint; a plain copy when k1 == 0, as the reference does).
 * sd_image_bounds gives mnMinX, mnMaxX, mnMinY, mnMaxY for sd_camera. */
int sd_undistort_points_device(const float* d_pts, int n, const float* K4, const float* dist5, float* d_out, void* stream);
/* mvKeysUn as a second key-point array of the batch (cameras with Camera.k1 != 0: TUM1 / TUM2 / EuRoC settings).  After
 * sd_batch_set_distortion every consumer that reads mvKeysUn in the reference does so here: the grid (PosInGrid / GetFeaturesInArea),
 * both SearchByProjection overloads, ComputeStereoFromRGBD's `kpU.pt.x - mbf/d` (Frame.cc:1069), UnprojectStereo, the point pairs of
 * TrackHomo and classifyH / classifyF (mvdynKeysUn); box membership, the depth lookup and the stereo matcher keep reading mvKeys.
 * sd_batch_undistort (re)computes the arrays of the listed slots -- call it after an extraction, after sd_batch_first_separate
 * (it permutes mvKeys) and after sd_batch_update_frame (it appends); sd_tracker_* does.  dist5[0] == 0 switches it off again. */
int sd_batch_set_distortion(sd_batch* b, const float* K4, const float* dist5);
int sd_batch_undistort(sd_batch* b, int n_slots, const int32_t* slots, void* stream);
int sd_batch_download_keys_un(sd_batch* b, int image, sd_keypoint* kp, int cap, int* n);
int sd_batch_download_dynamic_keys_un(sd_batch* b, int slot, sd_keypoint* kp, int cap, int* n); /* mvdynKeysUn, indexed like sd_batch_download_dynamic */
int sd_batch_undistort_keypoints(sd_batch* b, int n_images, const float* K4, const float* dist5, sd_keypoint* d_keys_un, void* stream);
int sd_image_bounds(int cols, int rows, const float* K4, const float* dist5, float* bounds4);

/* Frame copy constructor (src/Frame.cc:39-63), as in `mLastFrame = Frame(mCurrentFrame)`: copies the frame
 * results of slot src (keypoints, descriptors, mvuRight/mvDepth, grid cells, map-point table) to slot dst. */
int sd_batch_copy_frame(sd_batch* b, int src, int dst, void* stream);
int sd_batch_matches_device(sd_batch* b, int32_t** d_match, int32_t** d_pairs, int32_t** d_npairs, int32_t** d_nmatches,
                            int* cap);
int sd_batch_download_matches(sd_batch* b, int pair, int32_t* match, int32_t* pairs, int cap, int* npairs, int* nmatches);

/* ---- dynamic-object handling (src/Frame.cc:481-641, src/Tracking.cc:1093-1367) ---- */
/* Capacity of every per-frame box table of this ABI (the reference's vector<cv::Rect2d> is unbounded, include/Frame.h:55-68): a frame
 * may bring up to SD_MAX_BOXES detector boxes, and objects (the boxes after Frame::boxTrack's re-injection of unmatched last-frame boxes)
 * must fit the same table -- SD_ERR_CAPACITY says so explicitly when it does not; nothing is ever truncated. */
#define SD_MAX_BOXES 64
/* Frame::boxTrack(boxes, last_frame) (src/Frame.cc:481-552), host code (f64, a handful of boxes).  boxes: [cap][4]
 * (x, y, width, height) in/out — unmatched last-frame boxes are re-injected once, so *n_out may exceed n_box.
 * Outputs box_idx / omit / velocity ([cap], [cap], [cap][2]) are the frame's members of the same names. */
int sd_box_track(double* boxes, int n_box, int cap, const double* last_objects, int n_last, const int32_t* last_box_idx,
                 const uint8_t* last_omit, const double* last_velocity, int img_cols, int img_rows, int32_t* box_idx,
                 uint8_t* omit, double* velocity, int* n_out);
/* Frame::firstSeparate + the ctor split (src/Frame.cc:555-604, 336-367) for n_frames frame slots: keypoints
 * inside any box become dynamic (class_id = pre-split index), arrays are reordered static-first, N becomes
 * N_s, empty boxes are erased exactly as the reference does it (quirk included), per-box keypoint lists
 * (mvdynKeys/mdynDescriptors/mvudynRight/mvdynDepth) are index lists into the frame's arrays.
 * boxes: [n_frames][SD_MAX_BOXES][4], box_idx: [n_frames][SD_MAX_BOXES] (boxTrack outputs).  Run after the
 * stereo / RGB-D step (the lookup is per keypoint, so the order relative to the reorder is immaterial).
 * BOUND (this library's, the reference has none -- Frame.cc:555-604 works on std::vectors): the dynamic-object kernels keep a frame's key-point masks
 * in LDS (20 bytes per key-point slot): on a workspace whose extractor yields more than about 6,800 key points per image this call and
 * sd_batch_separate return SD_ERR_UNSUPPORTED (extraction and matching are not affected).  Rounds 1-3 stopped at 2,048; since round 4 the tables are sized by the workspace and a box's descriptors pass
 * through LDS in chunks of 2,048 (tests/test_gpu_cull.py: 5,000 features, one box holding 4,600 of them).  Nothing is ever truncated silently. */
int sd_batch_first_separate(sd_batch* b, int n_frames, const int32_t* slots, const double* boxes, const int32_t* n_boxes,
                            const int32_t* box_idx, void* stream);
/* Frame::objects / box_idx / box_status and the per-box lists of a slot.  kept_orig[j] = index (before the
 * erase) of surviving box j — apply the same selection to omit / box_velocity on the host.  box_start has nb+1
 * entries; box_items index the frame's DYNAMIC keypoint arrays (sd_batch_download_dynamic): the keypoints that
 * fell inside boxes, in original order — mvdynKeys[b][k] == dynamic[box_items[box_start[b] + k]]. */
int sd_batch_download_boxes(sd_batch* b, int slot, int* nb, double* boxes, int32_t* box_idx, int32_t* box_status,
                            int32_t* kept_orig, int32_t* box_start, int32_t* box_items, int items_cap, int* n_all,
                            int* n_static);
/* The per-slot record behind sd_batch_download_boxes as it lies in HBM ([max_images] of them): what a multi-GPU job gathers
 * beside the keypoint arrays (objects, box_idx, box_status, N_s = n_static, N_d = n_dynamic). */
typedef struct sd_frame_boxes {
    int32_t nb, n_all, n_static, n_dynamic;
    double boxes[SD_MAX_BOXES][4];
    int32_t box_idx[SD_MAX_BOXES], box_status[SD_MAX_BOXES], kept_orig[SD_MAX_BOXES];
    int32_t box_start[SD_MAX_BOXES + 1];
    int32_t pad_;
} sd_frame_boxes;
int sd_batch_boxes_device(sd_batch* b, sd_frame_boxes** d_frame_boxes);
/* The dynamic keypoints of a slot (class_id = index before the split), their descriptors, mvuRight and mvDepth. */
int sd_batch_download_dynamic(sd_batch* b, int slot, sd_keypoint* kp, uint8_t* desc, float* uright, float* depth, int cap, int* n);
/* Tracking::Separate(HorF, flag, dynStatus) (src/Tracking.cc:1093-1239) for n_pairs (current, reference) slots:
 * per box with the same id in both frames cv::BFMatcher(NORM_HAMMING, crossCheck).match, the <3 / <20 % skip,
 * classifyH (flag 1, :1241-1309) or classifyF (flag 2, :1311-1367) with H/F row-major 3x3 f32, the static /
 * dynamic box decision and box_status update against mLastFrame's (last_box_idx / last_box_status, [n_pairs][SD_MAX_BOXES]).
 * HorF == flag == NULL: pair p takes both from pair p of the preceding sd_batch_estimate_motion without leaving the
 * device; a pair whose TrackHomo flag is 0 is skipped as Tracking::Track_new does (ret 0, nothing classified). */
int sd_batch_separate(sd_batch* b, int n_pairs, const int32_t* cur_index, const int32_t* ref_index, const float* HorF,
                      const int32_t* flag, const int32_t* last_box_idx, const int32_t* last_box_status,
                      const int32_t* n_last, void* stream);
/* ret = Separate's return value; dyn_start[SD_MAX_BOXES + 1] / dyn_status = dynStatus as CSR over the current frame's boxes
 * (entries: index into the box list or -1); matches = (queryIdx, trainIdx) per entry. */
int sd_batch_download_separate(sd_batch* b, int pair, int32_t* ret, int32_t* dyn_start, int32_t* dyn_status, int32_t* matches,
                               int cap);
/* Frame::UpdateFrame(dynStatus) (src/Frame.cc:607-641) on the current frames of the last sd_batch_separate:
 * re-admitted keypoints are appended behind the static ones (class_id de-duplicated, push_back order), N grows.
 * only_if_static != 0 applies it only where Separate returned 1 (Tracking.cc:651-653).  Re-run
 * sd_batch_assign_grid afterwards (UpdateFeaturesToGrid). */
int sd_batch_update_frame(sd_batch* b, int only_if_static, void* stream);

/* Reference-frame queue of the dynamic block of Tracking::Track_new (src/Tracking.cc:620-666, 952-959; q_frame,
 * Tracking.h:109): host logic.  candidate(): oldest queued frame more than 0.2 s older than the current one
 * that has boxes (box-less fronts are dropped), or -1; reject(): TrackHomo failed on it, pop unless last;
 * push(): after a tracked frame, keep at most 0.3*fps frames (max_frames = Camera.fps). */
typedef struct sd_refqueue sd_refqueue;
int sd_refqueue_create(sd_refqueue** out);
int sd_refqueue_destroy(sd_refqueue* q);
int sd_refqueue_clear(sd_refqueue* q);
int sd_refqueue_size(const sd_refqueue* q, int* n);
int sd_refqueue_candidate(sd_refqueue* q, double cur_timestamp, int cur_has_boxes, int* slot);
int sd_refqueue_reject(sd_refqueue* q, int* again);
int sd_refqueue_push(sd_refqueue* q, double timestamp, int slot, int has_boxes, int max_frames, int* evicted_slot);

/* ---- Frame-level boundary: Tracking::GrabImage* -> Frame::Frame -> the dynamic block of Tracking::Track_new ----
 * sd_tracker is the per-frame front end of ORB_SLAM2::Tracking for n_lanes independent camera streams ("lanes": one
 * System each in the reference) that advance together, one frame per lane per call.  One call =
 *   System::TrackStereo / TrackRGBD / TrackMonocular (include/System.h:66-79, src/System.cc:119-375)
 *   -> Tracking::GrabImageStereo / GrabImageRGBD / GrabImageMonocular (src/Tracking.cc:170-343): cvtColor, depth scaling
 *   -> Frame::Frame, all five constructors (src/Frame.cc:66-126, 129-237, 240-294, 297-403, 406-461): extract (both eyes),
 *      boxTrack, firstSeparate + split, stereo / RGB-D association, grid
 *   -> Track_new's dynamic block (src/Tracking.cc:620-666): reference frame from q_frame, TrackHomo (:968-1086, th / 2*th
 *      retry, H / F fit), Separate (:1093-1239), UpdateFrame (Frame.cc:607-641)
 *   -> [track_last] TrackWithMotionModel's SearchByProjection(mCurrentFrame, mLastFrame, th) (src/Tracking.cc:1714-1741)
 *   -> q_frame.push / mLastFrame = Frame(mCurrentFrame) (src/Tracking.cc:952-959).
 * What is NOT behind this boundary is the SLAM state: the pose prediction mVelocity * mLastFrame.mTcw is an input (Tcw,
 * identity when NULL), a frame's map points are its own stereo points (Frame::UnprojectStereo), and `mState == OK &&
 * !mVelocity.empty()` is taken to hold from a lane's third frame on.  Stereo + boxes follows DESIGN.md Q9.
 * Results stay in the tracker's sd_batch (sd_tracker_batch): lane s's mCurrentFrame is slot cur_slot (valid until the next
 * call), TrackHomo's pair index is s, the last-frame matcher's pair index is n_lanes + s (sd_batch_download_matches /
 * _motion / _separate); every sd_batch_download_* call works on these slots. */
#define SD_SENSOR_MONOCULAR 0 /* System::eSensor, include/System.h:60-64 */
#define SD_SENSOR_STEREO 1
#define SD_SENSOR_RGBD 2
typedef struct sd_tracker sd_tracker;
typedef struct sd_tracker_params {
    int32_t sensor;         /* SD_SENSOR_* */
    int32_t width, height;
    int32_t channels;       /* input images: 1 = 8-bit gray, 3 = 8-bit BGR / RGB, 4 = BGRA / RGBA (cvtColor fused into the level-0 copy;
                             * Tracking.cc:175-200) */
    int32_t rgb_order;      /* Camera.RGB (Tracking.cc:107-112) */
    int32_t n_lanes;        /* independent camera streams */
    int32_t track_last;     /* != 0: also match against mLastFrame */
    int32_t depth_type;     /* RGB-D: element type of the depth image, SD_DEPTH_U16 (CV_16U) or SD_DEPTH_F32 (CV_32F: passed through when
                             * DepthMapFactor is 1, scaled in f32 otherwise -- Tracking.cc:271-272) */
    sd_camera cam;          /* Camera.fx .. Camera.bf and the image bounds */
    float dist[5];          /* Camera.k1, k2, p1, p2, k3 (k1 == 0: mvKeysUn == mvKeys, as Frame.cc:814-818) */
    float fps;              /* Camera.fps = mMaxFrames (Tracking.cc:93-98) */
    float depth_map_factor; /* DepthMapFactor (Tracking.cc:141-146); RGB-D only */
    float th_depth;         /* ThDepth (kept for the caller; not used on this path) */
    int32_t ini_features;   /* monocular: nFeatures of mpIniORBextractor (Tracking.cc:127-128: 2 * nFeatures), used while a lane is not
                             * initialised (Tracking.cc:335-338); 0 = one extractor for every frame */
    int32_t lookahead;      /* > 0: frames per lane sd_tracker_prefetch may extract ahead in one call (two such blocks can be outstanding) */
} sd_tracker_params;
#define SD_DEPTH_U16 0
#define SD_DEPTH_F32 1
typedef struct sd_lane_result {
    int32_t frame_id;       /* mnId within the lane (0, 1, ...) */
    int32_t cur_slot;       /* sd_batch slot of mCurrentFrame */
    int32_t last_slot;      /* slot of the copy kept as mLastFrame and as q_frame's newest entry */
    int32_t ref_slot, ref_frame_id; /* the reference frame TrackHomo ran against last (-1: the dynamic block did not run) */
    int32_t track_flag;     /* TrackHomo's return value: 0 failed / not run, 1 H, 2 F */
    int32_t separate_ret;   /* Separate's return value (0 when it did not run) */
    int32_t n_track_matches, n_track_pairs, n_h, n_f; /* nmatches, points_last.size(), n_H, n_F of TrackHomo */
    int32_t n_last_matches; /* SearchByProjection(mCurrentFrame, mLastFrame) or -1 */
    int32_t N, N_s, N_d;    /* keypoints after UpdateFrame; static / dynamic after the ctor's split */
    int32_t n_boxes;        /* objects.size() after the empty-box erase */
    int32_t box_idx[SD_MAX_BOXES], box_status[SD_MAX_BOXES];
    uint8_t omit[SD_MAX_BOXES];
    uint8_t pad_[4];
    double objects[SD_MAX_BOXES][4], box_velocity[SD_MAX_BOXES][2];
} sd_lane_result;
int sd_tracker_create(sd_tracker** out, sd_extractor* ex, const sd_tracker_params* params);
int sd_tracker_destroy(sd_tracker* t);
int sd_tracker_reset(sd_tracker* t);              /* Tracking::Reset (src/Tracking.cc:2369-2409): every lane starts over */
int sd_tracker_batch(sd_tracker* t, sd_batch** b); /* the workspace that holds the frames (owned by the tracker) */
/* One frame of every lane.  d_images: image k (0 = left / the only one, 1 = right) of lane s starts at
 * d_images + (s * images_per_lane + k) * image_pitch, rows `stride` bytes apart, `channels` bytes per pixel.
 * d_depth (RGB-D): depth image (CV_16U, or CV_32F when params.depth_type says so) of lane s at d_depth + s * depth_pitch_elems elements.  boxes [n_lanes][SD_MAX_BOXES][4]
 * (x, y, w, h) / n_boxes [n_lanes]: the detector's boxes of this frame; n_boxes[s] < 0 (or boxes == NULL) selects the
 * constructor without boxes for that lane.  timestamps [n_lanes].  Tcw / Twc (nullable, [n_lanes][16] row-major): pose
 * of the frame and its inverse.  results [n_lanes] (nullable).  The call returns after the results are on the host. */
int sd_tracker_track(sd_tracker* t, const uint8_t* d_images, size_t stride, size_t image_pitch, const void* d_depth,
                     size_t depth_stride_elems, size_t depth_pitch_elems, const double* boxes, const int32_t* n_boxes,
                     const double* timestamps, const float* Tcw, const float* Twc, sd_lane_result* results, void* stream);
/* The same from host images (a per-frame caller such as System::TrackStereo): images[s * images_per_lane + k] points to image k
 * of lane s in host memory (rows `stride` bytes apart), depth[s] to lane s's CV_16U depth image (RGB-D).  Uploads, then
 * sd_tracker_track. */
int sd_tracker_track_host(sd_tracker* t, const uint8_t* const* images, size_t stride, const void* const* depth,
                          size_t depth_stride_elems, const double* boxes, const int32_t* n_boxes, const double* timestamps,
                          const float* Tcw, const float* Twc, sd_lane_result* results);
/* Time-batched mode (BASELINE configs[4]: whole sequences, few lanes per GPU).  A frame's work splits into a history-free part --
 * GrabImage*'s cvtColor, ORB extraction of both eyes, UndistortKeyPoints, ComputeStereoMatches / ComputeStereoFromRGBD: more than 95 % of
 * the front end's time -- and the recurrence along the stream (boxTrack -> firstSeparate -> TrackHomo vs the queued frame -> Separate ->
 * UpdateFrame -> match vs mLastFrame), whose box ids (`max + 1`, Frame.cc:545-550) depend on the stream's whole history, so a stream
 * cannot be cut into independently processed chunks.  sd_tracker_prefetch runs the history-free part for the NEXT n_frames frames of every
 * lane in one batch (image e of frame k of lane s at d_images + ((k * n_lanes + s) * images_per_lane + e) * image_pitch, depth image of
 * frame k of lane s at d_depth + (k * n_lanes + s) * depth_pitch_elems), asynchronously on `stream`; the following n_frames calls of
 * sd_tracker_track with d_images == NULL consume them in order and run only the recurrence.  Results are identical to n_frames plain
 * calls.  Needs params.lookahead >= n_frames; two blocks may be outstanding (extract block b + 1 while block b is tracked). */
int sd_tracker_prefetch(sd_tracker* t, const uint8_t* d_images, size_t stride, size_t image_pitch, const void* d_depth,
                        size_t depth_stride_elems, size_t depth_pitch_elems, int n_frames, void* stream);
/* Frames -- not streams -- sharded over GPUs (BASELINE configs[4] on N > 1 ranks: "independent frames shard across the 8 GPUs").  Everything
 * a frame computes before Frame::boxTrack (src/Frame.cc:129-161: GrabImage*'s cvtColor, ORB extraction of both eyes, UndistortKeyPoints,
 * ComputeStereoMatches / ComputeStereoFromRGBD) depends on the frame's images alone; the recurrence (Frame.cc:162 onward, Tracking.cc:620-666,
 * 952-959) stays with the GPU that owns the stream.  So ANY GPU may run sd_tracker_prefetch on any frames (a "worker" tracker whose lanes are
 * simply batch entries), export the results as fixed-stride records, move them (RCCL all-to-all, the caller's business) and the owner imports
 * them as a prefetched block that sd_tracker_track(d_images == NULL) consumes exactly like a block it had extracted itself.
 *   record = N, per-level counts, mvKeys, mDescriptors, mvuRight, mvDepth, stereo SADs [, mvKeysUn when Camera.k1 != 0], each padded to 16 bytes:
 *            sd_tracker_prefetched_record_bytes(t) bytes (0 for a tracker without lookahead), the same for every tracker built with the same
 *            extractor parameters, image size and distortion setting -- the pyramid is not part of it (nothing after the association reads it).
 *   export : frames [first_frame, first_frame + n_frames) of the NEWEST outstanding prefetched block -> d_records[(k * n_lanes + s)] in
 *            frame-major order, on `stream` (waits for the block's extraction).  The caller orders a later prefetch into the same block behind it
 *            (same stream or an event).
 *   import : n_frames * n_lanes records in frame-major order become the next outstanding block (params.lookahead >= n_frames; two blocks may be
 *            outstanding); d_records may be reused once `stream` has passed this call.
 *   discard: a worker drops its newest block after exporting it (it will never track it).
 *   record_stride >= the record size, a multiple of 16 (d_records 16-byte aligned): the caller may keep its own bytes behind a record -- the
 *            detector's boxes of the frame travel there in bench.py. */
size_t sd_tracker_prefetched_record_bytes(const sd_tracker* t);
int sd_tracker_export_prefetched(sd_tracker* t, int first_frame, int n_frames, void* d_records, size_t record_stride, void* stream);
int sd_tracker_import_prefetched(sd_tracker* t, const void* d_records, size_t record_stride, int n_frames, void* stream);
int sd_tracker_discard_prefetched(sd_tracker* t);
/* The SLAM state a caller with a live back end owns, handed over the boundary (all optional; without them the tracker runs in its
 * sharded batch mode):
 *  - the pose prior is the Tcw / Twc argument of sd_tracker_track: `mCurrentFrame.SetPose(mVelocity*mLastFrame.mTcw)` (Tracking.cc:982)
 *    and the mRwc / mOw of Frame::UpdatePoseMatrices (Frame.cc:663-675);
 *  - sd_tracker_set_mappoints: the MapPoints of the frame just tracked, as the back end left them (mCurrentFrame.mvpMapPoints after
 *    TrackWithMotionModel / TrackLocalMap): xw [n_lanes][cap][3] = mvpMapPoints[i]->GetWorldPos(), flags [n_lanes][cap] (bit0: the point
 *    exists and is not an outlier, bit1: Observations() > 0), n [n_lanes] = entries given for the lane (its N), < 0 = leave the lane's
 *    own stereo points.  Call it after sd_tracker_track: it replaces the map-point table of the lane's mLastFrame, which is also the
 *    newest q_frame entry (they are copies of the same frame, Tracking.cc:952-959), i.e. what TrackHomo's and TrackWithMotionModel's
 *    SearchByProjection project in later frames (Tracking.cc:998-1010, ORBmatcher.cc:1485-1627).  The tracker never rewrites it;
 *  - sd_tracker_set_state: [n_lanes] bit0 = the lane is initialised (mState != NOT_INITIALIZED / NO_IMAGES_YET: the monocular lane uses
 *    mpORBextractorLeft instead of mpIniORBextractor, Tracking.cc:335-338), bit1 = `mState==OK && !mVelocity.empty()` (TrackHomo may
 *    run, Tracking.cc:971).  NULL returns to the automatic rule (frame 0, 1 of a lane: neither; later: both). */
int sd_tracker_set_mappoints(sd_tracker* t, const float* xw, const uint8_t* flags, const int32_t* n);
int sd_tracker_set_state(sd_tracker* t, const int32_t* state);
/* Batch copy of frame slots (Frame's copy constructor, src/Frame.cc:39-63) in one launch: slot src[i] -> dst[i]. */
int sd_batch_copy_frames(sd_batch* b, int n, const int32_t* src, const int32_t* dst, void* stream);

/* ---- dense RGB-D back-projection with the dynamic mask ----
 * PointCloudMapping::generatePointCloud(kf, color, depth, mask, dyn_obj) (src/pointcloudmapping.cc:59-103) for n_frames frame
 * slots: every third row / column; a pixel is skipped when it lies inside a dynamic box AND mask != 0, or when its depth is
 * outside [0.01, 5]; dyn_obj = the slot's objects whose box_status is 0 or 2 (Tracking::CreateNewKeyFrame, src/Tracking.cc:
 * 1999-2007).  d_color: 8-bit 3-channel image as handed to TrackRGBD (bytes 0, 1, 2 of a pixel become b, g, r); d_depth: CV_16U
 * with depth_factor = 1 / DepthMapFactor (the imDepth of Tracking.cc:271-272); d_mask: the 8-bit mask image (0 = background;
 * the reference converts it to CV_32F and tests != 0), may be NULL (nothing is masked).  Twc_host: [n_frames][16] row-major f64,
 * the T.inverse().matrix() of the key frame pose.  Output: d_points [n_frames][cap_points] in the reference's push_back order
 * (pcl::PointXYZRGBA payload), d_counts [n_frames][2] = {cloud size, masked_num}.  cap_points >= ceil(W/3) * ceil(H/3). */
typedef struct sd_cloud_point { float x, y, z; uint8_t b, g, r, a; } sd_cloud_point;
int sd_batch_backproject_dense(sd_batch* b, int n_frames, const int32_t* slots, const uint8_t* d_color, size_t color_stride,
                               size_t color_pitch, const uint16_t* d_depth, size_t depth_stride_elems, size_t depth_pitch_elems,
                               float depth_factor, const uint8_t* d_mask, size_t mask_stride, size_t mask_pitch, const sd_camera* cam,
                               const double* Twc_host, sd_cloud_point* d_points, int cap_points, int32_t* d_counts, void* stream);

/* ---- detector: yolov3Segment (include/yolo.h:22-48, src/yolo.cc, src/yolo/yolov3.cfg) ----
 * cv::dnn's Darknet importer + Net::forward + the reference's post-processing, on MFMA (f16 operands, f32
 * accumulation).  The network is given as a layer list (the five layer types of yolov3.cfg); weights are the
 * payload of a Darknet .weights file (floats after the header; per convolutional layer: biases, [scales, rolling
 * mean, rolling variance], weights [filters][c][size][size]), i.e. what readNetFromDarknet (yolo.cc:27) consumes. */
#define SD_YOLO_CONV 0
#define SD_YOLO_SHORTCUT 1
#define SD_YOLO_ROUTE 2
#define SD_YOLO_UPSAMPLE 3
#define SD_YOLO_YOLO 4
typedef struct sd_yolo_layer {
    int32_t type;
    int32_t filters, size, stride, batch_normalize, leaky; /* [convolutional]; leaky 0 = linear; pad = size/2 */
    int32_t from[2], nfrom;                                /* [shortcut] from / [route] layers (negative = relative) */
    int32_t mask[3];                                       /* [yolo] anchor indices */
} sd_yolo_layer;
typedef struct sd_yolo sd_yolo;
/* The 107 layers and 9 anchors of src/yolo/yolov3.cfg. */
int sd_yolo_v3_layers(sd_yolo_layer* layers, int cap, int* n, float anchors[18]);
/* yolov3Segment::yolov3Segment (yolo.cc:15-31); net_w x net_h = inpWidth x inpHeight (yolo.h:26-27). */
int sd_yolo_create(sd_yolo** out, const sd_yolo_layer* layers, int n_layers, const float anchors[18], int classes, int net_w,
                   int net_h, int max_batch);
/* The same with the arithmetic chosen: SD_YOLO_F16 = f16 operands, f32 accumulation on v_mfma_f32_32x32x16_f16 (default, the
 * throughput mode); SD_YOLO_F32 = f32 operands and accumulation on v_mfma_f32_32x32x2_f32, i.e. the reference's own arithmetic
 * (cv::dnn computes in f32, src/yolo.cc:29), at 1/16 of the MFMA rate.  In F32 mode sd_yolo_download_layer returns floats.
 * SD_YOLO_F32W = SD_YOLO_F32 with the 3 x 3 stride-1 layers of >= 64 input channels and >= 128 filters computed as Winograd
 * F(2 x 2, 3 x 3) in f32 (2.25 x fewer multiplies on those layers; sums of inputs and of weights are multiplied, so the last bits
 * differ from the direct f32 convolution -- held to the same layer tolerance and box-set test as SD_YOLO_F32). */
#define SD_YOLO_F16 0
#define SD_YOLO_F32 1
#define SD_YOLO_F32W 2
/* SD_YOLO_F32X3 = SD_YOLO_F32 with the >= 64-filter layers computed on three bf16 limbs per f32 operand (x = hi + mid + lo exactly;
 * a product = its six limb products of weight >= 2^-16, each exact in f32, accumulated in f32 by v_mfma_f32_32x32x16_bf16): what is
 * dropped is <= 2^-23 of a product, the size of an f32 multiply's own rounding.  Held to the same layer tolerance and box-set test. */
#define SD_YOLO_F32X3 3
int sd_yolo_create_prec(sd_yolo** out, const sd_yolo_layer* layers, int n_layers, const float anchors[18], int classes, int net_w,
                        int net_h, int max_batch, int precision);
int sd_yolo_precision(const sd_yolo* y, int* precision);
int sd_yolo_destroy(sd_yolo* y);
int sd_yolo_weight_count(const sd_yolo* y, size_t* n_floats);
int sd_yolo_load_darknet_weights(sd_yolo* y, const float* payload, size_t n_floats);
int sd_yolo_layer_shape(const sd_yolo* y, int layer, int* h, int* w, int* c);
int sd_yolo_flops(const sd_yolo* y, double* flops_per_image);
/* The MFMA FLOPs the chosen mode actually executes per image (== sd_yolo_flops except in SD_YOLO_F32W). */
int sd_yolo_mfma_flops(const sd_yolo* y, double* flops_per_image);
/* bf16 MFMA FLOPs per image executed by the limb kernels of SD_YOLO_F32X3 (0 in the other modes; sd_yolo_mfma_flops then counts
 * only the layers that stay on the f32 MFMA). */
int sd_yolo_mfma_flops_bf16(const sd_yolo* y, double* flops_per_image);
/* How many convolutions the mode computes as Winograd F(2 x 2, 3 x 3) (0 except in SD_YOLO_F32W). */
int sd_yolo_winograd_layers(const sd_yolo* y, int* n_layers);
/* blobFromImage + net.forward + the confidence filter (yolo.cc:63-68,163-183) for n 8-bit 3-channel images in HBM
 * (channel order as cv::imread delivers it, i.e. BGR; swapRB is applied as in the reference). */
int sd_yolo_forward_device(sd_yolo* y, const uint8_t* d_bgr, int width, int height, size_t stride, size_t image_pitch, int n,
                           float conf_threshold, void* stream);
/* Test access: layer output (f16, NHWC, dense) of one image; region-layer rows [total_rows][5 + classes] of image 0
 * (only after a forward with n == 1). */
int sd_yolo_download_layer(sd_yolo* y, int layer, int image, uint16_t* f16_nhwc_out); /* float* in SD_YOLO_F32 mode */
int sd_yolo_download_region(sd_yolo* y, float* rows, int* total_rows);
/* Host-image forms for a per-frame caller (`yolo->Segmentation_(imLeft)`, stereo_kitti.cc:107): upload + forward of one
 * 8-bit BGR image, and yolov3Segment::Segmentation's mask in host memory (image 0 of the last forward). */
int sd_yolo_forward_host(sd_yolo* y, const uint8_t* bgr, int width, int height, size_t stride, float conf_threshold);
int sd_yolo_mask_host(sd_yolo* y, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold, uint8_t* mask,
                      size_t stride, int* no_target);
/* yolov3Segment::Segmentation_ result for one image (yolo.cc:151-206): NMSBoxes(conf, nms), class filter
 * {person, car, bicycle, bus, truck}, box width -20 % / height +60 % about the centre.  boxes: [cap][4] x,y,w,h. */
/* yolov3Segment::Segmentation (yolo.cc:34-58, postprocess :80-137): d_mask (frame_rows x frame_cols u8 in HBM) = 1 except
 * inside the 31x31-ellipse dilation of the kept boxes' central halves; all ones and *no_target = 1 when none is kept. */
int sd_yolo_mask_device(sd_yolo* y, int image, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold,
                        uint8_t* d_mask, size_t stride, int* no_target, void* stream);
int sd_yolo_boxes(sd_yolo* y, int image, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold, double* boxes,
                  int32_t* class_ids, float* confidences, int cap, int* n);
/* The same post-processing for the first n_images of the last forward pass on the device (sort by score, greedy NMS,
 * class filter, rectCenterScale): boxes [n_images][SD_MAX_BOXES][4] f64 (x, y, w, h), class ids / confidences
 * [n_images][SD_MAX_BOXES], n_boxes[n_images].  _device writes device buffers and does not synchronise; _batch also
 * downloads (one synchronisation for the whole batch).  More than 4096 rows above the threshold or more than
 * SD_MAX_BOXES kept boxes in an image -> SD_ERR_CAPACITY (use the per-image host form). */
/* Overlap mode (f32-class modes; off by default).  A pass is blobFromImage -> 75 convolutions -> three region decodes -> (caller) NMS + download; on ONE
 * stream the small kernels at both ends (about 1.5 % of a 256-image pass) serialise with the next pass's convolutions.  With overlap on,
 * sd_yolo_forward_device(stream) issues the convolutions on `stream`, blobFromImage on an internal stream ahead of them and the decodes on another
 * behind their heads (events order them; consecutive passes are protected against each other the same way), and sd_yolo_boxes_device / _batch wait for
 * the decodes on THEIR stream argument -- give them a stream of their own and enqueue the next pass before consuming this one's boxes
 * (bench.py: two passes ahead).  `stream` of sd_yolo_forward_device is then NOT a completion point for the decoded rows; the host forms
 * (sd_yolo_boxes, sd_yolo_mask_*) synchronise the device as before.  Results are the same bits either way (tests/test_gpu_yolo.py). */
int sd_yolo_set_overlap(sd_yolo* y, int on);
int sd_yolo_boxes_device(sd_yolo* y, int n_images, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold,
                         double* d_boxes, int32_t* d_class_ids, float* d_confidences, int32_t* d_n_boxes, void* stream);
int sd_yolo_boxes_batch(sd_yolo* y, int n_images, int frame_cols, int frame_rows, float conf_threshold, float nms_threshold,
                        double* boxes, int32_t* class_ids, float* confidences, int32_t* n_boxes, void* stream);

/* ---- Tracking::GrabImage* preprocessing (src/Tracking.cc:170-343) ---- */
/* cvtColor(RGB|BGR|RGBA|BGRA -> GRAY); rgb_order = Camera.RGB.  channels 3 or 4. */
int sd_cvt_gray_device(const uint8_t* d_src, int width, int height, size_t src_stride, size_t src_pitch, int channels,
                       int rgb_order, uint8_t* d_dst, size_t dst_stride, size_t dst_pitch, int n_images, void* stream);
/* imDepth.convertTo(CV_32F, factor) for CV_16U input. */
int sd_depth_to_f32_device(const uint16_t* d_src, int width, int height, size_t src_stride_elems, float factor,
                           float* d_dst, int n_images, size_t src_pitch_elems, void* stream);

/* ---- ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1804-1820) ---- */
int sd_descriptor_distance(const uint8_t a[32], const uint8_t b[32]); /* host, returns 0..256 */
/* All-pairs Hamming distances of two descriptor sets in HBM: out[i*nb + j] (u16). */
int sd_hamming_matrix_device(const uint8_t* d_a, int na, const uint8_t* d_b, int nb, uint16_t* d_out, void* stream);

/* ---- profiling support for bench.py ----
 * When enabled, every kernel of the batch pipeline is bracketed by hipEvents on the stream it is
 * launched on; sd_batch_kernel_times returns the accumulated milliseconds and launch counts. */
int sd_batch_set_profiling(sd_batch* b, int enabled);
int sd_batch_kernel_count(const sd_batch* b, int* n);
int sd_batch_kernel_times(sd_batch* b, int index, const char** name, double* total_ms, int64_t* launches);
int sd_batch_reset_kernel_times(sd_batch* b);
int sd_batch_sync(sd_batch* b);

#ifdef __cplusplus
}
#endif
#endif /* SD_FRONTEND_H */
