// Internal declarations shared by the translation units of libvo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/vo_hip.h"

namespace vo {

constexpr int kLevels = 8;
constexpr int kEdge = 31;        // ORB edgeThreshold
constexpr int kHalfPatch = 15;   // patchSize / 2
constexpr int kFastThr = 20;
constexpr int kSelCap = 8192;    // largest per-level list the select kernel sorts in LDS

// Per-level geometry of a batch of equally sized images.
struct LevelGeom {
    int w, h;
    int pitch;           // bytes per row in the pyramid / blur blocks (level 0 of pyr: input step)
    uint32_t off;        // byte offset of the level inside one image's pyramid / blur block
    float scale;         // layerScale
    int quota;           // nfeaturesPerLevel
    int cand_off;        // offset (entries) of the level's candidate list inside one image's block
    int cand_cap;
    int kp_off;          // offset (entries) of the level's selected-keypoint list
    int kp_cap;
    int fast_tx, fast_ty, fast_tile0;   // FAST tile grid of the level and first flat tile id
    int fast_ns;         // tiles of the level that are sampled at the plain threshold (all of them on a small level)
    int fast_u0;         // position in the FAST tile table of the level's first unsampled tile (they are contiguous)
    int blur_tx, blur_ty, blur_tile0;
    uint32_t xtab, ytab; // offsets into the resize coefficient table (levels >= 1)
    uint32_t gtab;       // offset of the per-column-group table (12 dwords per four output columns)
    uint32_t ncg, ncg_magic;   // column groups per row and ceil(2^32 / ncg)
};

struct Geom {
    LevelGeom L[kLevels];
    int rows, cols, step;
    int n_images;
    size_t in_stride;     // bytes between input images
    size_t blk_stride;    // bytes of one image's pyramid (and blur) block
    int cand_stride;      // candidate entries per image
    int kp_stride;        // selected-keypoint entries per image
    int fast_tiles, blur_tiles;
    // The FAST tile table lists level 0 first, then the other levels; inside each of the two
    // groups the sampled tiles come first: fast_s0 of level 0's, fast_s1 of the other levels'.
    int fast_s0, fast_s1;
    // per-tile (level << 24 | tile row << 12 | tile column) tables inside the resize table buffer,
    // and floor(2^32 / tiles) for the workgroup -> (image, tile) split
    uint32_t fast_ttab, blur_ttab, fast_magic, blur_magic;
    int max_features;
    int sel_cap;          // LDS list capacity chosen for the select kernel
};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};

}  // namespace vo

// What one batch's tail (ratio test, counts, RANSAC rounds, pose recovery) owns: its stream, the
// events that order it against the main stream, the kNN table and match lists it reads and the
// RANSAC workspace.  A context has two such sets and consecutive pipeline calls alternate between
// them (vo_ctx::swap_tail), so that the tails of two batches - chains of small dependent launches
// that leave most of the GPU idle - run side by side.  The fields live directly in vo_ctx (the
// set in use) and in vo_ctx::alt (the other one).
#define VO_TAIL_BUFS(X)                                                                                        \
    X(st_idx) X(st_dist) X(m_top2) X(m_qidx) X(m_tidx) X(m_dist) X(m_p1) X(m_p2) X(m_cnt) X(r_state) X(r_samples) \
    X(r_models) X(r_nmodels) X(r_models2) X(r_nmodels2) X(r_retry) X(r_roots) X(r_nroots) X(r_counts) X(r_best)   \
    X(r_mask) X(r_pts) X(r_ptsf)
struct vo_tail_set {
    hipStream_t side = nullptr;
    hipEvent_t ev_init = nullptr, ev_rdone = nullptr, ev_tail = nullptr;
    bool side_busy = false, tail_pending = false;
    const void* res_begin = nullptr;   // result records the set's most recent tail writes
    const void* res_end = nullptr;
#define X(name) vo::DevBuf name;
    VO_TAIL_BUFS(X)
#undef X
};

struct vo_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    // The tail of the batched pipeline (ratio test, RANSAC rounds, pose recovery) runs on a second
    // stream so that the next batch's detection (main stream) overlaps it; ev_init orders side
    // after main, ev_rdone main after side before a main-stream call reuses the match buffers or
    // the RANSAC workspace.
    hipStream_t side = nullptr;
    hipEvent_t ev_init = nullptr, ev_rdone = nullptr;
    bool side_busy = false;
    // The pipeline's tail (ratio test, counts, RANSAC) runs on the side stream and still reads the
    // keypoint records and the kNN table of its batch: ev_tail marks the end of those reads, and
    // the next kernel on the main stream that overwrites them waits for it (join_tail).
    hipEvent_t ev_tail = nullptr;
    // Per-call completion marks for vo_pipeline_wait: ticket t uses slot t % VO_TICKET_RING;
    // ev_in: the call's last read of its input images, ev_out: its last result record.
    unsigned long long call_seq = 0;
    hipEvent_t ev_in[VO_TICKET_RING] = {}, ev_out[VO_TICKET_RING] = {};
    bool tail_pending = false;
    // FAST on level 0 needs only the input images: it runs on a third stream beside the resize
    // chain that builds the other levels (ev_fork orders it after the counter reset, ev_fast0
    // orders candidate selection after it).
    const void* res_begin = nullptr;   // result records the most recent tail on `side` writes
    const void* res_end = nullptr;
    vo_tail_set alt;                   // the other tail set (see vo_tail_set)
    void swap_tail() {
        std::swap(side, alt.side);
        std::swap(ev_init, alt.ev_init);
        std::swap(ev_rdone, alt.ev_rdone);
        std::swap(ev_tail, alt.ev_tail);
        std::swap(side_busy, alt.side_busy);
        std::swap(tail_pending, alt.tail_pending);
        std::swap(res_begin, alt.res_begin);
        std::swap(res_end, alt.res_end);
#define X(name) std::swap(name, alt.name);
        VO_TAIL_BUFS(X)
#undef X
    }
    hipStream_t aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_fast0 = nullptr, ev_blur = nullptr;
    std::string err;
    int opt[VO_OPT_COUNT] = {};   // VO_OPT_* values (defaults set by vo_ctx_create)
    unsigned timing = 0;   // bit g: kernel group g is bracketed by events
    std::vector<hipEvent_t> ev_pool;                 // every event ever created (reused after a reset)
    size_t ev_used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pairs[VO_T_COUNT];

    // detection workspace
    vo::Geom g{};
    bool geom_valid = false;
    std::vector<uint32_t> h_tab;   // resize tables (host copy)
    vo::DevBuf tab, pyr, blur, cand, cand_cnt, kp_x, kp_y, kp_resp, kp_cnt, sel_keys, sel_resp, sel_cnt;
    // Truncation bits (VO_TRUNC_*): `flags` is one sticky word per context that every kernel able
    // to truncate a list ORs into and that only take_truncation() clears; `img_status` holds one
    // word per image of the most recent detection call.
    vo::DevBuf flags, img_status;
    // Word of `flags` the detection kernels OR into: 0, the sticky word, for device-resident calls
    // (which can only report through vo_sync); 16, a word nobody reads, for host-buffer calls,
    // which wait for their own result and report from their images' status words instead.
    int flags_word = 0;
    vo::DevBuf fast_stats;                // four 64-bit scene counters (vo_debug_fast_stats)
    vo::DevBuf fast_thr, fast_redo;       // raised FAST threshold per (image, level); the two launch groups' redo lists
    bool fast_raised = false;             // the most recent detection call ran with raised thresholds
    bool fast_stats_on = false;
    uint32_t* img_status_cur = nullptr;   // the half of img_status the most recent detection call uses
    int status_phase = 0;
    const uint8_t* last_imgs = nullptr;
    // staging for the host-buffer entry points
    vo::DevBuf st_img, st_kps, st_desc, st_cnt, st_q, st_t, st_idx, st_dist, st_down;
    void* h_pin = nullptr;       // pinned host staging of vo_match_images (upload and download block)
    size_t h_pin_bytes = 0;
    // matching / RANSAC workspace
    vo::DevBuf m_top2, m_qidx, m_tidx, m_dist, m_p1, m_p2, m_cnt;
    vo::DevBuf k_part;   // per-chunk two best of a Hamming search split over the train set (main stream only)
    vo::DevBuf r_state, r_samples, r_models, r_nmodels, r_models2, r_nmodels2, r_retry, r_roots, r_nroots, r_counts, r_best, r_mask, r_pts, r_ptsf, r_res;
    // pipeline workspace
    vo::DevBuf p_kps, p_desc, p_cnt;
    // VO_OPT_PIPELINE_GRAPH: the launch chain of the most recent repeated pipeline call as a HIP
    // graph.  graph_key: the call it was captured from (arguments, options, allocation count);
    // graph_seen: the last call that ran the ordinary way (a call is captured when it comes twice
    // in a row); alloc_count: workspace (re)allocations so far - a graph holds device addresses,
    // so any of them retires it.
    struct PipelineKey {
        const void* imgs;
        void* results;
        size_t image_stride;
        double K[4], threshold, confidence;
        float ratio;
        int n_images, n_pairs, pair_step, rows, cols, step, max_features, fixed_iters;
        int opt[VO_OPT_COUNT];
        unsigned long long alloc_count;
    };
    bool ransac_short_hint = false;   // the last RANSAC call from host points ended within its first round
    PipelineKey graph_key{}, graph_seen{};
    hipGraphExec_t graph_exec = nullptr;
    unsigned long long alloc_count = 0;
};

namespace vo {

int fail(vo_ctx* c, int code, const std::string& msg);
// Records the begin / end event of one launch of kernel group `group` (no-ops unless timing is on).
void t_begin(vo_ctx* c, int group);
void t_end(vo_ctx* c, int group);
void t_begin_on(vo_ctx* c, int group, hipStream_t s);
void t_end_on(vo_ctx* c, int group, hipStream_t s);   // events on another stream of the context
// Makes the main stream wait for work still running on the side stream.
int join_side(vo_ctx* c);
// Makes the main stream wait until the side stream has finished reading the previous batch's
// keypoints and kNN table.
int join_tail(vo_ctx* c);
int ensure(vo_ctx* c, DevBuf& b, size_t bytes);
// Reads and clears the context's sticky truncation word (the streams must have been waited for):
// VO_OK if it was zero, else VO_ERR_CAPACITY with the stages named in the error message.
int take_truncation(vo_ctx* c, const char* who);
// VO_ERR_CAPACITY with the stages of the VO_TRUNC_* bits named in the error message.
int truncation_error(vo_ctx* c, const char* who, uint32_t bits);

#define VO_HIP(c, call)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return vo::fail((c), VO_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// orb
int orb_prepare(vo_ctx* c, int n_images, int rows, int cols, int step, size_t in_stride, int max_features);
int orb_run(vo_ctx* c, const uint8_t* d_imgs, vo_keypoint* d_kps, uint8_t* d_desc, int cap, int32_t* d_counts);
// match
int knn2_launch(vo_ctx* c, const uint8_t* d_q, const int32_t* d_nq, int cnt_stride, int nq_max, size_t q_stride, const uint8_t* d_t,
                const int32_t* d_nt, int nt_max, size_t t_stride, int batch, int32_t* d_idx2, int32_t* d_dist2,
                int nq_typical = 0,    // expected query count when nq_max is only a capacity (picks the workgroup size)
                hipStream_t on = nullptr);   // a tail stream to run on instead of the main stream (see vo_pipeline.hip)
int ratio_launch(vo_ctx* c, const int32_t* d_idx2, const int32_t* d_dist2, const int32_t* d_nq, int cnt_stride,
                 int nq_max, int batch,
                 float ratio, const vo_keypoint* d_kp1, const vo_keypoint* d_kp2, size_t kp_stride, int32_t* d_qidx,
                 int32_t* d_tidx, float* d_dist, float* d_p1, float* d_p2, int32_t* d_nmatch, hipStream_t s);
// Resolves, once per GPU, the device address of the static workspace of vo_match_ratio_static
// (called by vo_ctx_create with the context's device current).
int static_scratch_init(vo_ctx* c);
// ransac
int ransac_dev(vo_ctx* c, const float* d_p1, const float* d_p2, const int32_t* d_n, int n_max, int batch,
               const double K[4], double threshold, double confidence, int fixed_iters, vo_pair_result* d_res,
               double* d_E, uint8_t* d_mask_e, uint8_t* d_mask_pose, int32_t* d_good, int32_t* d_samples,
               int samples_cap, bool on_side, bool ask_after_first_round = false);

}  // namespace vo
