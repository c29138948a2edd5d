/* C ABI of the MI355X (gfx950) visual-odometry front end.
 *
 * The reference has no FFI seam of its own: its hot path is three free C++
 * functions compiled into one static library.  This header is the boundary a
 * replacement of those functions binds to; each entry point names the
 * reference interface it stands in for.  Plain pointers and sizes only,
 * caller-owned buffers, int status codes (0 = success).
 *
 * Pointers named d_* are device (HBM) addresses on the context's GPU; all
 * others are host addresses.  Device entry points enqueue work on the
 * context's stream and return without waiting; vo_sync() waits.  Host entry
 * points copy in, run, copy out and return when the results are in place.
 * A context is not thread-safe: use one per host thread and GPU.
 */
#ifndef VO_HIP_H
#define VO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vo_ctx vo_ctx;

/* Same 28-byte layout as cv::KeyPoint (pt.x, pt.y, size, angle, response,
 * octave, class_id) as filled by cv::ORB. */
typedef struct vo_keypoint {
    float x, y;
    float size;
    float angle;    /* degrees in [0,360) */
    float response; /* Harris response */
    int32_t octave;
    int32_t class_id; /* always -1 */
} vo_keypoint;

/* One frame pair's result of the fused pipeline (motion_estimate plus the
 * counts the reference's benchmark reports). */
typedef struct vo_pair_result {
    double R[9]; /* row-major rotation */
    double t[3]; /* unit translation */
    int32_t inliers;
    int32_t valid;
    int32_t n_kp1, n_kp2;
    int32_t n_matches;
    int32_t ransac_iters;
    int32_t status;   /* VO_TRUNC_* bits of the pair's two images; 0 = complete */
    int32_t models_scored; /* essential-matrix candidates of the ransac_iters hypotheses (each is tested on every match) */
} vo_pair_result;

/* A keypoint list of an image did not fit an internal table or the caller's
 * capacity and was cut (deterministically: the entries first in raster order
 * are kept).  OpenCV's retainBest keeps every candidate tied with the last one
 * retained, so on images with thousands of equal scores the lists are
 * unbounded; the tables hold 2 x quota + 512 (at least 2048) candidates per
 * level after the FAST-score cut and quota + 256 keypoints per level. */
enum {
    VO_TRUNC_SCORE_LIST = 1,    /* ties at the FAST-score cut (2 x level quota) */
    VO_TRUNC_KEYPOINT_LIST = 2, /* ties at the Harris-response cut (level quota) */
    VO_TRUNC_OUTPUT = 4         /* more keypoints than the output capacity `cap` */
};

enum {
    VO_OK = 0,
    VO_ERR_INVALID = 1,  /* bad argument */
    VO_ERR_HIP = 2,      /* a HIP runtime call failed */
    VO_ERR_CAPACITY = 3, /* an output or internal list overflowed its capacity */
    VO_ERR_NOMEM = 4
};

/* --- context ------------------------------------------------------------- */
/* Number of GPUs a context can be created on (0 if the HIP runtime finds none). */
int vo_device_count(void);
int vo_ctx_create(int device, vo_ctx** out);
void vo_ctx_destroy(vo_ctx* ctx);
/* Message of the last failing call on this context ("" if none). */
const char* vo_last_error(const vo_ctx* ctx);
/* hipStream_t the context enqueues on (as void*).  Detection forks part of its
 * work (FAST on level 0) to an internal auxiliary stream and joins it again
 * before candidate selection, so work enqueued on this stream after a
 * detection call is ordered after all of it.  vo_pipeline_*_dev also use a
 * second internal stream that is NOT joined - see there. */
void* vo_ctx_stream(vo_ctx* ctx);
/* Waits for every stream of the context.  Returns VO_ERR_CAPACITY (message:
 * which stage) if any detection enqueued since the last report truncated a
 * keypoint list - device-resident calls cannot report that themselves, they
 * only enqueue.  The condition is reported once and then cleared. */
int vo_sync(vo_ctx* ctx);
/* Kernel timing with HIP events recorded on the stream a kernel group runs on.
 * vo_set_timing(ctx, groups) takes a bit mask (1 << VO_T_*, VO_T_ALL for every
 * group, 0 to switch timing off): every launch of a selected kernel group is
 * then bracketed by a fresh event pair.  Nothing waits on the events, but each
 * bracket costs about 10 us of stream time, so select only what is needed
 * inside a throughput measurement.  vo_kernel_time_ms() waits for the stream,
 * then returns the mean duration in milliseconds over the launches recorded
 * since the last vo_set_timing() and their number in *launches; -1 if none
 * (FAST has two launches per detection call: level 0 and the upper levels).
 * While the resize group is selected, detection keeps all its kernels on one
 * stream, so that a group's time is that of its kernels alone; otherwise the
 * level-0 FAST launch shares the GPU with the resize chain.
 * Groups: */
enum {
    VO_T_RESIZE = 0,   /* the seven pyramid resize launches of one detect call */
    VO_T_FAST = 1,     /* FAST-9 + NMS + border filter, all levels */
    VO_T_SELECT = 2,   /* score / Harris selection + orientation */
    VO_T_BLUR = 3,     /* 7x7 blur, all levels */
    VO_T_DESCRIBE = 4, /* rBRIEF + keypoint assembly */
    VO_T_KNN2 = 5,     /* Hamming top-2 */
    VO_T_RATIO = 6,    /* ratio test + compaction */
    VO_T_RANSAC = 7,   /* all RANSAC and pose kernels of one call */
    /* inside the RANSAC chain, per round (selecting one of these splits the chain's bracket;
     * do not combine them with VO_T_RANSAC in one measurement): */
    VO_T_RANSAC_SOLVE = 8, /* five-point solves: solve_front + solve_back */
    VO_T_RANSAC_SCORE = 9, /* Sampson scoring of every (hypothesis, model) over the points */
    VO_T_COUNT = 10,
    VO_T_ALL = (1 << 8) - 1 /* the eight top-level groups */
};
int vo_set_timing(vo_ctx* ctx, int groups);
float vo_kernel_time_ms(vo_ctx* ctx, int group, int* launches);

/* Per-context switches (no counterpart in the reference, whose functions take
 * only the config structs of ref:include/visual_odometry/feature_detector.hpp:24-26,
 * feature_matcher.hpp:26-28, motion_estimator.hpp:48-51).  None of them changes a
 * result except VO_OPT_FIVE_POINT_LITERAL; the others pick between kernels that
 * must agree bit for bit, for tests and measurements.  A context takes its
 * defaults once, in vo_ctx_create, from the environment variable named with
 * each option (unset: the value marked default); after that only
 * vo_ctx_set_option changes them.  Options are read when a call enqueues work,
 * so set them between calls, from the thread that owns the context. */
enum {
    /* 1: the minimal solver as cv::findEssentialMat runs it (one chart of the
     * null space, the roots as they come, no refinement, no duplicate removal,
     * no second chart); 0 (default): complete, refined solution sets.
     * Environment: VO_FIVE_POINT_LITERAL. */
    VO_OPT_FIVE_POINT_LITERAL = 0,
    /* FAST with per-(image, level) raised thresholds: -1 (default) decided by
     * the batch size, 0 never, 1 always.  Environment: VO_FAST_RAISE. */
    VO_OPT_FAST_RAISE = 1,
    /* Hamming search kernel: -1 (default) and 2 the matrix-core kernel, 0 the
     * scalar-operand vector kernel, 1 the LDS-broadcast vector kernel (which
     * hands train sets beyond 65 536 rows with very large grids to the
     * scalar-operand one), 3 the matrix-core kernel with four query groups per
     * wavefront whatever the size (test aid).  All give the same table.
     * Environment: VO_KNN_MODE. */
    VO_OPT_KNN_MODE = 2,
    /* Wavefronts per workgroup of the matrix-core and LDS-broadcast kernels: 0
     * (default) by size, else 1, 4, 8 or 16.  Environment: VO_KNN_NW. */
    VO_OPT_KNN_WAVES = 3,
    /* Queries per lane of the scalar-operand kernel: 0 (default) by size, else
     * 1, 2 or 4.  Environment: VO_KNN_QPL. */
    VO_OPT_KNN_QUERIES_PER_LANE = 4,
    /* 0: skip the second-chart pass of the five-point solver (measurement aid;
     * changes results); 1 (default).  Environment: VO_SECOND_CHART. */
    VO_OPT_SECOND_CHART = 5,
    /* Train-set chunks a single Hamming search is split into (grid = query
     * blocks x chunks, merged by a second kernel): 0 (default) by size, 1 never,
     * else 2..64.  Environment: VO_KNN_SPLIT. */
    VO_OPT_KNN_TRAIN_SPLIT = 6,
    /* Root kernel of the five-point solver: 0 (default) one wavefront per eight
     * hypotheses, 1 one 512-lane workgroup per 64 (tuning aid; the roots do not
     * depend on it).  Environment: VO_SOLVE_VARIANT. */
    VO_OPT_SOLVE_VARIANT = 7,
    /* 1: a batched pipeline call (vo_pipeline_pairs_dev / vo_pipeline_sequence_dev)
     * that repeats the previous call's arguments exactly - same buffers, shape and
     * parameters, as in a serving loop - is captured as a HIP graph on the context's
     * stream and replayed from then on: one graph launch instead of the chain of
     * some fifty kernel launches over three streams, which is what a call on a few
     * pairs consists of.  The replay runs everything on the one stream, so it does
     * not overlap a batch's RANSAC rounds with the next batch's detection: meant for
     * small batches.  Same results.  0 (default): off.  Environment:
     * VO_PIPELINE_GRAPH. */
    VO_OPT_PIPELINE_GRAPH = 8,
    VO_OPT_COUNT = 9
};
int vo_ctx_set_option(vo_ctx* ctx, int option, int value);
int vo_ctx_get_option(const vo_ctx* ctx, int option, int* value);

/* --- detect_features / detect_keypoints_only -------------------------------
 * Replaces visual_odometry::detect_features (ref:src/feature_detector.cpp:12-18,
 * decl ref:include/visual_odometry/feature_detector.hpp:36) and, with
 * desc_out == NULL, detect_keypoints_only (ref:src/feature_detector.cpp:20-27):
 * ORB with nfeatures = max_features and OpenCV's other defaults (scale 1.2,
 * 8 levels, edge 31, FAST 20, Harris score, patch 31, WTA_K 2).
 * img: rows x cols 8-bit grayscale, `step` bytes per row.  Keypoints come out
 * per pyramid level, in raster order inside a level.  rows*cols == 0 yields
 * n_out = 0.  More than `cap` keypoints -> VO_ERR_CAPACITY. */
int vo_orb_detect(vo_ctx* ctx, const uint8_t* img, int rows, int cols, int step, int max_features,
                  vo_keypoint* kps_out, uint8_t* desc_out /* cap x 32 or NULL */, int cap, int* n_out);

/* Batched, device-resident form of the same: n_images images of identical
 * shape, image i at d_imgs + i*image_stride.  Image i's keypoints land at
 * d_kps + i*cap, descriptors at d_desc + i*cap*32 (d_desc may be NULL), its
 * count at d_counts[i]. */
int vo_orb_detect_batch_dev(vo_ctx* ctx, const uint8_t* d_imgs, int n_images, int rows, int cols,
                            int step, size_t image_stride, int max_features, vo_keypoint* d_kps,
                            uint8_t* d_desc, int cap, int32_t* d_counts);

/* --- match_features ---------------------------------------------------------
 * Replaces the cv::BFMatcher(NORM_HAMMING, crossCheck=false)::knnMatch(k=2)
 * call of visual_odometry::match_features (ref:src/feature_matcher.cpp:26-30):
 * for every 32-byte query row the two nearest train rows by Hamming distance,
 * nearest first, ties to the lower train index.  idx2/dist2 are nq x 2; a
 * missing neighbour (nt < 2) is idx -1, dist -1. */
int vo_hamming_knn2(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, int32_t* idx2,
                    int32_t* dist2);
int vo_hamming_knn2_dev(vo_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt,
                        int32_t* d_idx2, int32_t* d_dist2);

/* The same search over several GPUs of one node, for descriptor sets so large
 * that one search is worth sharding (the reference's knnMatch call,
 * ref:src/feature_matcher.cpp:26-30, is one serial scan): ctxs[k], one context
 * per GPU created by the caller, takes the query rows
 * vo_block_range(nq, k, n_ctx, ...); the train set is replicated; a query's two
 * nearest rows are complete inside its block, so nothing is reduced - the
 * blocks' rows are only brought together.  Results are those of one
 * vo_hamming_knn2 call.
 * vo_hamming_knn2_multi: host buffers; one host thread per context uploads,
 * searches and downloads into rows [begin, end) of idx2 / dist2; returns when
 * they are complete.
 * vo_hamming_knn2_multi_dev: d_q[k] = block k's query rows on ctxs[k]'s GPU,
 * d_t[k] = the whole train set on that GPU (16-byte aligned), d_idx2 / d_dist2
 * = the full nq x 2 tables on ctxs[0]'s GPU; every other context sends its rows
 * there with one peer copy per table on its own stream (the path's single
 * exchange step).  Only enqueues; complete after vo_sync() on every context. */
int vo_hamming_knn2_multi(vo_ctx* const* ctxs, int n_ctx, const uint8_t* q, int nq, const uint8_t* t, int nt,
                          int32_t* idx2, int32_t* dist2);
int vo_hamming_knn2_multi_dev(vo_ctx* const* ctxs, int n_ctx, const uint8_t* const* d_q, int nq,
                              const uint8_t* const* d_t, int nt, int32_t* d_idx2, int32_t* d_dist2);

/* The whole of match_features (ref:src/feature_matcher.cpp:15-42): kNN(2),
 * Lowe ratio test `d0 < ratio * d1` in float32, survivors in query order.
 * Outputs hold up to nq entries: query index, train index, distance (the
 * fields of cv::DMatch; imgIdx is always 0).  Empty inputs give *n_out = 0. */
int vo_match_ratio(vo_ctx* ctx, const uint8_t* q, int nq, const uint8_t* t, int nt, float ratio,
                   int32_t* query_idx, int32_t* train_idx, float* distance, int* n_out);

/* The device-resident form of vo_match_ratio for callers that cannot allocate
 * device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  It has no counterpart in the reference.  The kNN(2)
 * table between the search and the ratio test lives in a workspace that is part
 * of the library's code object (16 bytes per query row, VO_STATIC_MAX_QUERIES
 * rows = 512 KiB on every GPU the library runs on, held until the process ends),
 * so the call performs no hipMalloc / hipFree and only enqueues two kernels on
 * the context's stream.  d_q: nq x 32, d_t: nt x 32 (16-byte aligned); outputs
 * hold up to nq entries, *d_n_out the number written; all in device memory and
 * valid once the stream has run.  nq > VO_STATIC_MAX_QUERIES -> VO_ERR_CAPACITY.
 * All contexts of one GPU share that GPU's workspace: calls that overlap on one
 * GPU (two contexts, two host threads) race and must be serialised by the caller. */
#define VO_STATIC_MAX_QUERIES 32768
int vo_match_ratio_static(vo_ctx* ctx, const uint8_t* d_q, int nq, const uint8_t* d_t, int nt, float ratio,
                          int32_t* d_query_idx, int32_t* d_train_idx, float* d_distance, int32_t* d_n_out);

/* --- estimate_motion -----------------------------------------------------------
 * Replaces visual_odometry::estimate_motion (ref:src/motion_estimator.cpp:44-95):
 * cv::findEssentialMat(RANSAC, confidence, threshold, 1000 iterations, 5-point
 * solver, Sampson error) then cv::recoverPose (cheirality, depth < 50), with
 * the reference's validity rules (fewer than 5 points, no model, or fewer than
 * 10 pose inliers -> *valid = 0 and *inliers = 0).  p1/p2: n x 2 float pixel
 * coordinates.  K = {fx, fy, cx, cy}.  fixed_iters > 0 evaluates exactly that
 * many hypotheses instead of the adaptive count.  mask (n bytes, may be NULL)
 * receives the pose inlier mask. */
int vo_estimate_motion(vo_ctx* ctx, const float* p1, const float* p2, int n, const double K[4],
                       double threshold, double confidence, int fixed_iters, double R[9],
                       double t[3], int* inliers, int* valid, uint8_t* mask);

/* RANSAC stage alone, for parity checks under a shared hypothesis set:
 * returns the winning essential matrix, its inlier mask and count, the number
 * of iterations run, and (samples != NULL, max(1000, fixed_iters) x 5 ints) the
 * 5-point index sets in the order drawn. */
int vo_find_essential(vo_ctx* ctx, const float* p1, const float* p2, int n, const double K[4],
                      double threshold, double confidence, int fixed_iters, double E[9],
                      uint8_t* mask, int* good, int* iters_run, int32_t* samples);

/* --- fused pipeline --------------------------------------------------------------
 * detect x2 + match + estimate_motion for n_pairs frame pairs, i.e.
 * match_images_orb (ref:src/image_matcher.cpp:71-81) followed by estimate_motion
 * as the reference's main loop runs them (ref:src/main.cpp:162-170), with every
 * intermediate kept in HBM.  Pair p uses images 2p and 2p+1 of d_imgs.
 * d_results: n_pairs records in device memory.  The call only enqueues:
 * everything behind the kNN search (ratio test, RANSAC rounds, pose recovery:
 * the call's tail) runs on one of two further streams of the context, which
 * consecutive calls use alternately, so that the next call's detection and the
 * next call's tail overlap it; d_results is complete after vo_sync() (which
 * waits for every stream of the context; waiting on vo_ctx_stream() alone is
 * not enough).  Consecutive calls that write overlapping records are ordered
 * (the later call's records are what is left); give consecutive calls distinct
 * record buffers to let their tails run side by side. */
int vo_pipeline_pairs_dev(vo_ctx* ctx, const uint8_t* d_imgs, int n_pairs, int rows, int cols, int step,
                          size_t image_stride, int max_features, float ratio, const double K[4],
                          double threshold, double confidence, int fixed_iters,
                          vo_pair_result* d_results);

/* Finer waits than vo_sync() for a caller that streams batches from host
 * memory (no counterpart in the reference, whose loop is synchronous): every
 * vo_pipeline_pairs_dev / vo_pipeline_sequence_dev call gets a ticket (1, 2, ...
 * per context; vo_pipeline_ticket returns the last one, 0 before the first
 * call).  vo_pipeline_wait blocks the host until that call has read its input
 * images for the last time (VO_WAIT_INPUTS: the image buffer may be overwritten,
 * e.g. by the upload of the batch after next) or has written all its result
 * records (VO_WAIT_RESULTS) - without waiting for calls enqueued after it, so the
 * RANSAC rounds of one batch keep running beside the detection of the next.
 * Every other call shares a tail stream and those finish in order; the last
 * VO_TICKET_RING tickets are tracked one by one, an older one is waited for
 * through the younger call that took its slot (VO_TICKET_RING is even, so that
 * call ran on the same tail stream).  Unlike vo_sync() this does
 * not report truncated keypoint lists; vo_pair_result.status carries them. */
enum { VO_WAIT_INPUTS = 1, VO_WAIT_RESULTS = 2, VO_TICKET_RING = 4 };
unsigned long long vo_pipeline_ticket(vo_ctx* ctx);
int vo_pipeline_wait(vo_ctx* ctx, unsigned long long ticket, int what);

/* match_images_orb (ref:src/image_matcher.cpp:71-81; orb_matcher::match_images,
 * ref:include/visual_odometry/image_matcher.hpp:52-55) for ONE pair of host
 * images, what the reference's main loop (ref:src/main.cpp:162) and benchmark
 * (ref:benchmarks/benchmark_matchers.cpp:112) call per pair: detect both, match,
 * ratio test.  One upload of the two images, one download of both keypoint sets
 * (at most `cap` each) and the surviving matches (at most `cap`: query index,
 * train index, distance, in query order); descriptors stay in HBM.  Returns
 * when the outputs are written.  rows * cols == 0 gives all counts 0. */
int vo_match_images(vo_ctx* ctx, const uint8_t* img1, const uint8_t* img2, int rows, int cols, int step,
                    int max_features, float ratio, vo_keypoint* kps1, int* n_kps1, vo_keypoint* kps2, int* n_kps2,
                    int cap, int32_t* query_idx, int32_t* train_idx, float* distance, int* n_matches);

/* The same for a video sequence: n_frames consecutive frames in d_imgs give the
 * n_frames - 1 pairs (i, i+1), as the reference's main loop walks a directory
 * (ref:src/main.cpp:152-184, which detects every frame twice, once per pair it
 * belongs to).  Each frame is detected once and serves as the second image of
 * one pair and the first of the next; results are those of
 * vo_pipeline_pairs_dev on the same pairs. */
int vo_pipeline_sequence_dev(vo_ctx* ctx, const uint8_t* d_imgs, int n_frames, int rows, int cols, int step,
                             size_t image_stride, int max_features, float ratio, const double K[4],
                             double threshold, double confidence, int fixed_iters,
                             vo_pair_result* d_results);

/* Host-buffer form: n_frames images in host memory (row pitch `step`, image
 * pitch image_stride), n_frames - 1 records out; returns when they are written.
 * What a directory-walking caller (ref:src/main.cpp:152-184) feeds a batch at a
 * time, re-sending the last frame of one batch as the first of the next. */
int vo_pipeline_sequence(vo_ctx* ctx, const uint8_t* imgs, int n_frames, int rows, int cols, int step,
                         size_t image_stride, int max_features, float ratio, const double K[4],
                         double threshold, double confidence, int fixed_iters, vo_pair_result* results);

/* The same over several GPUs of one node (the reference is a serial loop over
 * independent pairs, ref:src/main.cpp:152-184): ctxs[k], one context per GPU
 * created by the caller, takes the contiguous block of pairs
 * vo_block_range(n_frames - 1, k, n_ctx, ...) on a host thread of its own - the
 * frames of that block plus one, the boundary frame being detected by both
 * neighbours - and the records land at their pair index, so results are those
 * of one vo_pipeline_sequence call.  No data moves between GPUs.  Returns the
 * first hard error (message in ctxs[0]), VO_ERR_CAPACITY if only truncations
 * were reported. */
int vo_pipeline_sequence_multi(vo_ctx* const* ctxs, int n_ctx, const uint8_t* imgs, int n_frames, int rows, int cols,
                               int step, size_t image_stride, int max_features, float ratio, const double K[4],
                               double threshold, double confidence, int fixed_iters, vo_pair_result* results);
/* Contiguous block [*begin, *end) of n units owned by part `part` of `parts`
 * (sizes differ by at most one).  Host arithmetic, no GPU call. */
int vo_block_range(int n, int part, int parts, int* begin, int* end);

/* --- inspection (parity tests) ---------------------------------------------------
 * Copies pyramid level `level` of image `image` of the most recent detect call
 * (blurred != 0: the 7x7 Gaussian-blurred copy the descriptors sample) to
 * host memory, tightly packed w x h.  `out` holds out_bytes bytes: when that is
 * less than w * h, nothing is copied, *w and *h are still set and the call
 * returns VO_ERR_CAPACITY (out_bytes = 0 asks for the size). */
int vo_debug_level(vo_ctx* ctx, int image, int level, int blurred, uint8_t* out, size_t out_bytes, int* w, int* h);
/* FAST candidates (after non-max suppression and the 31-pixel border filter)
 * of one level of the most recent detect call as (x, y, score) triples, in no
 * particular order.  Returns the count in *n_out.  After a call that ran with
 * raised thresholds (vo_debug_fast_thresholds) the list holds every candidate
 * the selection can keep - all that score at least the level's threshold -
 * but only some of the weaker ones. */
int vo_debug_candidates(vo_ctx* ctx, int image, int level, int32_t* out, int cap, int* n_out);
/* Scene statistics of the FAST stage, for reporting what a benchmark scene looks
 * like: enable != 0 zeroes the counters and makes every following detection add
 * to them; enable == 0 waits for the streams, stops counting and returns
 * out[0] = pixels tested (the 31-px interior of every level, plus the 1-px ring whose
 * scores the non-max suppression needs), out[1] = pixels passing the four-cardinal
 * pre-test, out[2] = FAST-9 corners, out[3] = candidates left after non-max suppression
 * and the border filter. */
int vo_debug_fast_stats(vo_ctx* ctx, int enable, uint64_t out[4]);
/* What the FAST stage of the most recent detect call did on one image when it ran
 * with raised thresholds (batched calls; see fast_raise_kernel): out[l] = the
 * threshold the unsampled tiles of level l ran with (20 = not raised), out[8 + l]
 * = 1 if that level's tiles had to run again with the plain threshold.  *raised_out
 * = 0 (and out untouched) if the call ran every tile with the plain threshold. */
int vo_debug_fast_thresholds(vo_ctx* ctx, int image, int32_t out[16], int* raised_out);

/* The five-point minimal solver alone: `count` sets of 5 correspondences in
 * normalised image coordinates (x1, x2: count x 5 x 2 doubles).  Es receives
 * count x 10 x 9 doubles (unit-norm essential matrices, row-major, ascending
 * in the hidden variable), n_models the number found per set. */
int vo_debug_five_point(vo_ctx* ctx, const double* x1, const double* x2, int count, double* Es,
                        int32_t* n_models);

#ifdef __cplusplus
}
#endif
#endif /* VO_HIP_H */
