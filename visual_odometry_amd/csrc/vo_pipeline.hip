// Fused frame-pair pipeline: detect x2 -> kNN(2) + ratio -> E-RANSAC + pose,
// every intermediate resident in HBM.  Stands in for match_images_orb
// (ref:src/image_matcher.cpp:71-81) followed by estimate_motion as the
// reference's main loop chains them (ref:src/main.cpp:162-170).
#include <algorithm>
#include <cstring>
#include <thread>
#include <vector>

#include "vo_internal.h"

namespace vo {

// image of pair p: first = p * pair_step, second = first + 1 (pair_step 2: independent
// pairs, 1: consecutive frames of a sequence)
__global__ void fill_counts_kernel(vo_pair_result* __restrict__ res, const int32_t* __restrict__ kp_cnt,
                                   const int32_t* __restrict__ nmatch, const uint32_t* __restrict__ img_status,
                                   int n_pairs, int pair_step) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_pairs) return;
    res[p].n_kp1 = kp_cnt[pair_step * p];
    res[p].n_kp2 = kp_cnt[pair_step * p + 1];
    res[p].n_matches = nmatch[p];
    res[p].status = (int32_t)(img_status[pair_step * p] | img_status[pair_step * p + 1]);
}

// out[0] = OR of the n status words (n <= 64)
__global__ void or_status_kernel(const uint32_t* __restrict__ status, int n, uint32_t* __restrict__ out) {
    uint32_t v = (int)threadIdx.x < n ? status[threadIdx.x] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o);
    if (threadIdx.x == 0) out[0] = v;
}

}  // namespace vo

using namespace vo;

// Shared body: n_images frames, pair p = (p * pair_step, p * pair_step + 1).
static int pipeline_body(vo_ctx* c, const uint8_t* d_imgs, int n_images, int n_pairs, int pair_step, int rows, int cols,
                         int step, size_t image_stride, int max_features, float ratio, const double K[4],
                         double threshold, double confidence, int fixed_iters, vo_pair_result* d_results);

// ---- VO_OPT_PIPELINE_GRAPH: a repeated call as one graph launch
static void drop_graph(vo_ctx* c) {
    if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
    c->graph_exec = nullptr;
}

// Captures the call's launch chain on the main stream.  A capture cannot wait for events of work
// outside it and records no event that anything outside could wait for, so: everything earlier is
// waited for on the host first, the side and auxiliary streams are the main stream for the
// duration (their fork / join events then order a stream behind itself), and the events the chain
// touched are recorded again, outside the capture, before anyone looks at them.
static int capture_graph(vo_ctx* c, const uint8_t* d_imgs, int n_images, int n_pairs, int pair_step, int rows, int cols,
                         int step, size_t image_stride, int max_features, float ratio, const double K[4],
                         double threshold, double confidence, int fixed_iters, vo_pair_result* d_results) {
    drop_graph(c);
    VO_HIP(c, hipStreamSynchronize(c->stream));
    VO_HIP(c, hipStreamSynchronize(c->side));
    VO_HIP(c, hipStreamSynchronize(c->alt.side));
    VO_HIP(c, hipStreamSynchronize(c->aux));
    c->side_busy = c->alt.side_busy = false;
    c->tail_pending = c->alt.tail_pending = false;
    const hipStream_t side = c->side, aux = c->aux;
    const unsigned long long seq = c->call_seq, allocs = c->alloc_count;
    c->side = c->aux = c->stream;
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal);
    int rc = VO_OK;
    if (e == hipSuccess) {
        rc = pipeline_body(c, d_imgs, n_images, n_pairs, pair_step, rows, cols, step, image_stride, max_features, ratio, K,
                           threshold, confidence, fixed_iters, d_results);
        e = hipStreamEndCapture(c->stream, &graph);
    }
    c->side = side;
    c->aux = aux;
    c->call_seq = seq;   // (the replay takes the ticket)
    c->side_busy = c->tail_pending = false;
    if (rc == VO_OK && e == hipSuccess && c->alloc_count != allocs) {
        rc = fail(c, VO_ERR_HIP, "pipeline graph: the captured call allocated workspace");
    }
    if (rc == VO_OK && e == hipSuccess && graph) e = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    // the events the chain recorded inside the capture, made ordinary events again
    hipEvent_t touched[] = {c->ev_init, c->ev_rdone, c->ev_tail, c->ev_fork, c->ev_fast0, c->ev_blur,
                            c->ev_in[(seq + 1) % VO_TICKET_RING], c->ev_out[(seq + 1) % VO_TICKET_RING]};
    for (hipEvent_t ev : touched)
        if (ev) (void)hipEventRecord(ev, c->stream);
    if (rc != VO_OK) {
        drop_graph(c);
        return rc;
    }
    if (e != hipSuccess) {
        drop_graph(c);
        (void)hipGetLastError();
        return fail(c, VO_ERR_HIP, std::string("pipeline graph capture: ") + hipGetErrorString(e));
    }
    return VO_OK;
}

static int pipeline_run(vo_ctx* c, const uint8_t* d_imgs, int n_images, int n_pairs, int pair_step, int rows, int cols,
                        int step, size_t image_stride, int max_features, float ratio, const double K[4],
                        double threshold, double confidence, int fixed_iters, vo_pair_result* d_results) {
    VO_HIP(c, hipSetDevice(c->device));
    auto make_key = [&]() {
        vo_ctx::PipelineKey k;
        memset(&k, 0, sizeof(k));   // (the key is compared byte by byte, padding included)
        k.imgs = d_imgs;
        k.results = d_results;
        k.image_stride = image_stride;
        for (int i = 0; i < 4; ++i) k.K[i] = K[i];
        k.threshold = threshold;
        k.confidence = confidence;
        k.ratio = ratio;
        k.n_images = n_images; k.n_pairs = n_pairs; k.pair_step = pair_step; k.rows = rows; k.cols = cols; k.step = step;
        k.max_features = max_features; k.fixed_iters = fixed_iters;
        for (int i = 0; i < VO_OPT_COUNT; ++i) k.opt[i] = c->opt[i];
        k.alloc_count = c->alloc_count;
        return k;
    };
    if (c->opt[VO_OPT_PIPELINE_GRAPH] == 1 && !c->timing && !c->fast_stats_on) {
        const vo_ctx::PipelineKey k = make_key();
        if (!c->graph_exec || memcmp(&k, &c->graph_key, sizeof(k)) != 0) {
            if (memcmp(&k, &c->graph_seen, sizeof(k)) == 0) {   // the second call in a row with these arguments
                const int crc = capture_graph(c, d_imgs, n_images, n_pairs, pair_step, rows, cols, step, image_stride,
                                              max_features, ratio, K, threshold, confidence, fixed_iters, d_results);
                if (crc != VO_OK) return crc;
                c->graph_key = k;
            } else {
                drop_graph(c);
            }
        }
        if (c->graph_exec) {
            // the replay runs on the main stream, behind whatever earlier calls left on the others
            if (int jrc = join_side(c)) return jrc;
            if (int jrc = join_tail(c)) return jrc;
            VO_HIP(c, hipGraphLaunch(c->graph_exec, c->stream));
            const unsigned long long ticket = c->call_seq + 1;
            VO_HIP(c, hipEventRecord(c->ev_in[ticket % VO_TICKET_RING], c->stream));
            VO_HIP(c, hipEventRecord(c->ev_out[ticket % VO_TICKET_RING], c->stream));
            c->call_seq = ticket;
            return VO_OK;
        }
    } else if (c->graph_exec) {
        drop_graph(c);
    }
    // this call's tail takes the tail set the call before the previous one used: it runs beside
    // the previous call's tail (and behind the older one, in stream order).  A call that fails
    // takes no ticket, so it gives the set back: tickets and tail sets stay in step (a ticket's
    // parity names its tail stream, which vo_pipeline_wait relies on).
    c->swap_tail();
    const int rc = pipeline_body(c, d_imgs, n_images, n_pairs, pair_step, rows, cols, step, image_stride, max_features,
                                 ratio, K, threshold, confidence, fixed_iters, d_results);
    if (rc != VO_OK) c->swap_tail();
    if (rc == VO_OK && c->opt[VO_OPT_PIPELINE_GRAPH] == 1) {
        c->graph_seen = make_key();   // (after the call: what it allocated is there now)
    }
    return rc;
}

static int pipeline_body(vo_ctx* c, const uint8_t* d_imgs, int n_images, int n_pairs, int pair_step, int rows, int cols,
                         int step, size_t image_stride, int max_features, float ratio, const double K[4],
                         double threshold, double confidence, int fixed_iters, vo_pair_result* d_results) {
    int rc;
    if ((rc = orb_prepare(c, n_images, rows, cols, step, image_stride, max_features))) return rc;
    const int cap = c->g.kp_stride;   // quota + slack of every level: the detector cannot return more
    if ((rc = ensure(c, c->p_kps, (size_t)n_images * cap * sizeof(vo_keypoint)))) return rc;
    if ((rc = ensure(c, c->p_desc, (size_t)n_images * cap * 32))) return rc;
    if ((rc = ensure(c, c->p_cnt, (size_t)n_images * 4))) return rc;
    if ((rc = ensure(c, c->st_idx, (size_t)n_pairs * cap * 8))) return rc;
    if ((rc = ensure(c, c->st_dist, (size_t)n_pairs * cap * 8))) return rc;
    if ((rc = ensure(c, c->m_qidx, (size_t)n_pairs * cap * 4))) return rc;
    if ((rc = ensure(c, c->m_tidx, (size_t)n_pairs * cap * 4))) return rc;
    if ((rc = ensure(c, c->m_dist, (size_t)n_pairs * cap * 4))) return rc;
    if ((rc = ensure(c, c->m_p1, (size_t)n_pairs * cap * 8))) return rc;
    if ((rc = ensure(c, c->m_p2, (size_t)n_pairs * cap * 8))) return rc;
    if ((rc = ensure(c, c->m_cnt, (size_t)n_pairs * 4))) return rc;
    vo_keypoint* kps = (vo_keypoint*)c->p_kps.p;
    uint8_t* desc = (uint8_t*)c->p_desc.p;
    int32_t* cnt = (int32_t*)c->p_cnt.p;
    if ((rc = orb_run(c, d_imgs, kps, desc, cap, cnt))) return rc;
    // (the detector's last kernel follows every reader of the input images on this stream)
    const unsigned long long ticket = c->call_seq + 1;
    VO_HIP(c, hipEventRecord(c->ev_in[ticket % VO_TICKET_RING], c->stream));
    // Everything after the kNN search - ratio test, counts, RANSAC rounds, pose recovery - is a
    // chain of small, latency-bound kernels: it goes to the side stream, so that the main stream
    // can start the next batch's detection right behind the matcher.  On calls of middling size
    // the search goes there as well and runs beside the next batch's resize chain and first FAST
    // launch (measured in alternating runs on one box: 64 pairs of 2000 keypoints +3.7 %, 256
    // +2.5 %, 256 pairs of 4000 keypoints +1 %, 32 full-HD pairs +1.3 %).  Not on the smallest
    // (100-pair sequences of 1000 keypoints, a chain of short launches: -4 %) and not on the
    // largest: 512 dense pairs fill the GPU either way (+0.1 %) and only stretch the FAST launch
    // that the search then shares it with.  ev_tail marks the point after which the side stream
    // no longer reads this batch's keypoints, descriptors and kNN table; the next batch's
    // orientation kernel, the first to overwrite any of them, waits for it.
    hipStream_t side = c->side;
    const bool search_aside = (long long)n_pairs * max_features >= 128000 && n_pairs <= 256;
    const size_t dstride = (size_t)pair_step * cap * 32;
    if (!search_aside &&
        (rc = knn2_launch(c, desc, cnt, pair_step, cap, dstride, desc + (size_t)cap * 32, cnt + 1, cap, dstride, n_pairs,
                          (int32_t*)c->st_idx.p, (int32_t*)c->st_dist.p, max_features)))
        return rc;
    VO_HIP(c, hipEventRecord(c->ev_init, c->stream));
    VO_HIP(c, hipStreamWaitEvent(side, c->ev_init, 0));
    if (search_aside &&
        (rc = knn2_launch(c, desc, cnt, pair_step, cap, dstride, desc + (size_t)cap * 32, cnt + 1, cap, dstride, n_pairs,
                          (int32_t*)c->st_idx.p, (int32_t*)c->st_dist.p, max_features, side)))
        return rc;
    // a caller that hands consecutive calls the same result records gets them written in call
    // order: this tail then starts behind the previous call's, as it did with a single tail stream
    const uint8_t* const rb = (const uint8_t*)d_results;
    const uint8_t* const re = rb + (size_t)n_pairs * sizeof(vo_pair_result);
    if (c->alt.side_busy && rb < (const uint8_t*)c->alt.res_end && (const uint8_t*)c->alt.res_begin < re)
        VO_HIP(c, hipStreamWaitEvent(side, c->alt.ev_rdone, 0));
    c->res_begin = rb;
    c->res_end = re;
    if ((rc = ratio_launch(c, (const int32_t*)c->st_idx.p, (const int32_t*)c->st_dist.p, cnt, pair_step, cap, n_pairs,
                           ratio, kps, kps + cap, (size_t)pair_step * cap, (int32_t*)c->m_qidx.p, (int32_t*)c->m_tidx.p,
                           (float*)c->m_dist.p, (float*)c->m_p1.p, (float*)c->m_p2.p, (int32_t*)c->m_cnt.p, side)))
        return rc;
    hipLaunchKernelGGL(fill_counts_kernel, dim3((n_pairs + 63) / 64), dim3(64), 0, side, d_results, cnt,
                       (const int32_t*)c->m_cnt.p, (const uint32_t*)c->img_status_cur, n_pairs, pair_step);
    VO_HIP(c, hipEventRecord(c->ev_tail, side));
    c->tail_pending = true;
    if ((rc = ransac_dev(c, (const float*)c->m_p1.p, (const float*)c->m_p2.p, (const int32_t*)c->m_cnt.p, cap, n_pairs,
                         K, threshold, confidence, fixed_iters, d_results, nullptr, nullptr, nullptr, nullptr, nullptr,
                         0, true)))
        return rc;
    VO_HIP(c, hipEventRecord(c->ev_out[ticket % VO_TICKET_RING], side));
    c->call_seq = ticket;
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

extern "C" unsigned long long vo_pipeline_ticket(vo_ctx* c) { return c ? c->call_seq : 0ull; }

extern "C" int vo_pipeline_wait(vo_ctx* c, unsigned long long ticket, int what) {
    if (!c) return VO_ERR_INVALID;
    if (ticket == 0 || ticket > c->call_seq || (what & ~(VO_WAIT_INPUTS | VO_WAIT_RESULTS)) || what == 0)
        return fail(c, VO_ERR_INVALID, "vo_pipeline_wait: no such ticket or nothing to wait for");
    VO_HIP(c, hipSetDevice(c->device));
    // a slot that a younger call has taken over marks a later point of the same streams (the
    // calls alternate between two tail streams and the ring has an even number of slots)
    static_assert(VO_TICKET_RING % 2 == 0, "a ticket's slot is reused by calls on the same tail stream");
    if (what & VO_WAIT_INPUTS) VO_HIP(c, hipEventSynchronize(c->ev_in[ticket % VO_TICKET_RING]));
    if (what & VO_WAIT_RESULTS) VO_HIP(c, hipEventSynchronize(c->ev_out[ticket % VO_TICKET_RING]));
    return VO_OK;
}

extern "C" int vo_pipeline_pairs_dev(vo_ctx* c, const uint8_t* d_imgs, int n_pairs, int rows, int cols, int step,
                                     size_t image_stride, int max_features, float ratio, const double K[4],
                                     double threshold, double confidence, int fixed_iters,
                                     vo_pair_result* d_results) {
    if (!c) return VO_ERR_INVALID;
    if (n_pairs <= 0 || rows <= 0 || cols <= 0 || step < cols || max_features <= 0 || !d_imgs || !K || !d_results)
        return fail(c, VO_ERR_INVALID, "vo_pipeline_pairs_dev: bad argument");
    return pipeline_run(c, d_imgs, 2 * n_pairs, n_pairs, 2, rows, cols, step, image_stride, max_features, ratio, K,
                        threshold, confidence, fixed_iters, d_results);
}

extern "C" int vo_pipeline_sequence_dev(vo_ctx* c, const uint8_t* d_imgs, int n_frames, int rows, int cols, int step,
                                        size_t image_stride, int max_features, float ratio, const double K[4],
                                        double threshold, double confidence, int fixed_iters,
                                        vo_pair_result* d_results) {
    if (!c) return VO_ERR_INVALID;
    if (n_frames < 2 || rows <= 0 || cols <= 0 || step < cols || max_features <= 0 || !d_imgs || !K || !d_results)
        return fail(c, VO_ERR_INVALID, "vo_pipeline_sequence_dev: bad argument");
    return pipeline_run(c, d_imgs, n_frames, n_frames - 1, 1, rows, cols, step, image_stride, max_features, ratio, K,
                        threshold, confidence, fixed_iters, d_results);
}

// Host-buffer form of vo_pipeline_sequence_dev: frames in, records out, synchronous.
extern "C" int vo_pipeline_sequence(vo_ctx* c, const uint8_t* imgs, int n_frames, int rows, int cols, int step,
                                    size_t image_stride, int max_features, float ratio, const double K[4],
                                    double threshold, double confidence, int fixed_iters, vo_pair_result* results) {
    if (!c) return VO_ERR_INVALID;
    if (n_frames < 2 || rows <= 0 || cols <= 0 || step < cols || image_stride < (size_t)step * rows || max_features <= 0 ||
        !imgs || !K || !results)
        return fail(c, VO_ERR_INVALID, "vo_pipeline_sequence: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    int rc;
    const size_t img_bytes = (size_t)n_frames * image_stride, res_bytes = (size_t)(n_frames - 1) * sizeof(vo_pair_result);
    if ((rc = ensure(c, c->st_img, img_bytes))) return rc;
    if ((rc = ensure(c, c->r_res, res_bytes))) return rc;
    VO_HIP(c, hipMemcpyAsync(c->st_img.p, imgs, img_bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = pipeline_run(c, (const uint8_t*)c->st_img.p, n_frames, n_frames - 1, 1, rows, cols, step, image_stride,
                           max_features, ratio, K, threshold, confidence, fixed_iters, (vo_pair_result*)c->r_res.p)))
        return rc;
    rc = vo_sync(c);   // every stream: the records are complete
    if (rc != VO_OK && rc != VO_ERR_CAPACITY) return rc;
    // on a truncated keypoint list the records still go out (status names the pairs concerned)
    VO_HIP(c, hipMemcpy(results, c->r_res.p, res_bytes, hipMemcpyDeviceToHost));
    return rc;
}

// match_images_orb (ref:src/image_matcher.cpp:71-81) for one pair of host images in one round
// trip: both images go up in one copy, detection runs on the two-image batch, the matcher and
// the ratio test read the descriptors where the detector left them, and keypoints, matches and
// counts come back in one copy - no descriptor ever crosses the bus.  Pinned staging on both
// sides keeps the copies asynchronous; the call waits once.
extern "C" int vo_match_images(vo_ctx* c, const uint8_t* img1, const uint8_t* img2, int rows, int cols, int step,
                               int max_features, float ratio, vo_keypoint* kps1, int* n_kps1, vo_keypoint* kps2,
                               int* n_kps2, int cap, int32_t* query_idx, int32_t* train_idx, float* distance,
                               int* n_matches) {
    if (!c) return VO_ERR_INVALID;
    if (!n_kps1 || !n_kps2 || !n_matches) return fail(c, VO_ERR_INVALID, "vo_match_images: a count pointer is NULL");
    *n_kps1 = *n_kps2 = *n_matches = 0;
    if (!img1 || !img2 || rows <= 0 || cols <= 0) return VO_OK;   // an empty image has no features: no matches
    if (step < cols || cap <= 0 || max_features <= 0 || !kps1 || !kps2 || !query_idx || !train_idx || !distance)
        return fail(c, VO_ERR_INVALID, "vo_match_images: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    int rc;
    const size_t img_bytes = (size_t)rows * step;
    const size_t kp_bytes = (size_t)cap * sizeof(vo_keypoint);
    // download block: [n1, n2, n_matches, pad | kps1 | kps2 | qidx | tidx | dist]
    const size_t off_k1 = 16, off_k2 = off_k1 + kp_bytes, off_q = off_k2 + kp_bytes, off_t = off_q + (size_t)cap * 4,
                 off_d = off_t + (size_t)cap * 4, down_bytes = off_d + (size_t)cap * 4;
    const size_t pin_bytes = std::max(2 * img_bytes, down_bytes);
    if (c->h_pin_bytes < pin_bytes) {
        VO_HIP(c, hipStreamSynchronize(c->stream));
        if (c->h_pin) (void)hipHostFree(c->h_pin);
        c->h_pin = nullptr;
        c->h_pin_bytes = 0;
        VO_HIP(c, hipHostMalloc(&c->h_pin, pin_bytes, hipHostMallocDefault));
        c->h_pin_bytes = pin_bytes;
    }
    if ((rc = ensure(c, c->st_img, 2 * img_bytes))) return rc;
    if ((rc = ensure(c, c->st_down, down_bytes))) return rc;
    if ((rc = ensure(c, c->st_desc, 2 * (size_t)cap * 32))) return rc;
    if ((rc = ensure(c, c->st_idx, (size_t)cap * 8))) return rc;
    if ((rc = ensure(c, c->st_dist, (size_t)cap * 8))) return rc;
    if ((rc = join_side(c))) return rc;   // a pipeline call's tail may still use the staging and match buffers
    uint8_t* pin = (uint8_t*)c->h_pin;
    // (the last row of an image ends after `cols` bytes: a sub-image's buffer may end there)
    const size_t valid = (size_t)(rows - 1) * step + cols;
    std::memcpy(pin, img1, valid);
    std::memcpy(pin + img_bytes, img2, valid);
    VO_HIP(c, hipMemcpyAsync(c->st_img.p, pin, 2 * img_bytes, hipMemcpyHostToDevice, c->stream));
    if ((rc = orb_prepare(c, 2, rows, cols, step, img_bytes, max_features))) return rc;
    uint8_t* down = (uint8_t*)c->st_down.p;
    int32_t* d_cnt = (int32_t*)down;
    vo_keypoint* d_kps = (vo_keypoint*)(down + off_k1);
    uint8_t* d_desc = (uint8_t*)c->st_desc.p;
    c->flags_word = 16;   // this call reports from its own images' status words
    rc = orb_run(c, (const uint8_t*)c->st_img.p, d_kps, d_desc, cap, d_cnt);
    c->flags_word = 0;
    if (rc) return rc;
    if ((rc = knn2_launch(c, d_desc, d_cnt, 1, cap, 0, d_desc + (size_t)cap * 32, d_cnt + 1, cap, 0, 1, (int32_t*)c->st_idx.p,
                          (int32_t*)c->st_dist.p, max_features)))
        return rc;
    if ((rc = ratio_launch(c, (const int32_t*)c->st_idx.p, (const int32_t*)c->st_dist.p, d_cnt, 1, cap, 1, ratio, nullptr,
                           nullptr, 0, (int32_t*)(down + off_q), (int32_t*)(down + off_t), (float*)(down + off_d), nullptr,
                           nullptr, d_cnt + 2, c->stream)))
        return rc;
    // (the fourth word of the block's header takes the two images' truncation bits)
    hipLaunchKernelGGL(or_status_kernel, dim3(1), dim3(64), 0, c->stream, (const uint32_t*)c->img_status_cur, 2,
                       (uint32_t*)(d_cnt + 3));
    VO_HIP(c, hipMemcpyAsync(pin, down, down_bytes, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    const int32_t* h_cnt = (const int32_t*)pin;
    const int n1 = std::min(h_cnt[0], cap), n2 = std::min(h_cnt[1], cap), m = std::min(h_cnt[2], cap);
    const uint32_t status = (uint32_t)h_cnt[3];
    std::memcpy(kps1, pin + off_k1, (size_t)n1 * sizeof(vo_keypoint));
    std::memcpy(kps2, pin + off_k2, (size_t)n2 * sizeof(vo_keypoint));
    std::memcpy(query_idx, pin + off_q, (size_t)m * 4);
    std::memcpy(train_idx, pin + off_t, (size_t)m * 4);
    std::memcpy(distance, pin + off_d, (size_t)m * 4);
    *n_kps1 = n1;
    *n_kps2 = n2;
    *n_matches = m;
    // a truncated keypoint list is reported, but what was kept - and matched - is still handed out
    return status ? truncation_error(c, "vo_match_images", status) : VO_OK;
}

// Contiguous block [begin, end) of n units owned by part `part` of `parts` (sizes differ by at
// most one, larger blocks first).
extern "C" int vo_block_range(int n, int part, int parts, int* begin, int* end) {
    if (n < 0 || parts <= 0 || part < 0 || part >= parts || !begin || !end) return VO_ERR_INVALID;
    const int base = n / parts, rem = n % parts;
    *begin = part * base + std::min(part, rem);
    *end = *begin + base + (part < rem ? 1 : 0);
    return VO_OK;
}

// vo_pipeline_sequence over several GPUs of one node: the reference's loop over the pairs of a
// directory (ref:src/main.cpp:152-184) has independent iterations, so context k takes the
// contiguous block of pairs vo_block_range(n_frames - 1, k, n_ctx) - the frames of that block
// plus one, the frame shared with the next block being detected by both - on a host thread of
// its own, and writes its records at their pair index.  No device-to-device traffic.
extern "C" int vo_pipeline_sequence_multi(vo_ctx* const* ctxs, int n_ctx, const uint8_t* imgs, int n_frames, int rows,
                                          int cols, int step, size_t image_stride, int max_features, float ratio,
                                          const double K[4], double threshold, double confidence, int fixed_iters,
                                          vo_pair_result* results) {
    if (!ctxs || n_ctx <= 0) return VO_ERR_INVALID;
    for (int k = 0; k < n_ctx; ++k)
        if (!ctxs[k]) return VO_ERR_INVALID;
    if (n_frames < 2 || !imgs || !results || !K)
        return fail(ctxs[0], VO_ERR_INVALID, "vo_pipeline_sequence_multi: bad argument");
    for (int a = 0; a < n_ctx; ++a)
        for (int b = a + 1; b < n_ctx; ++b)
            if (ctxs[a] == ctxs[b])
                return fail(ctxs[0], VO_ERR_INVALID, "vo_pipeline_sequence_multi: a context is listed twice (contexts are not thread-safe)");
    const int n_pairs = n_frames - 1;
    std::vector<int> rc((size_t)n_ctx, VO_OK);
    std::vector<std::thread> workers;
    for (int k = 0; k < n_ctx; ++k) {
        int b = 0, e = 0;
        vo_block_range(n_pairs, k, n_ctx, &b, &e);
        if (e <= b) continue;   // more contexts than pairs
        workers.emplace_back([=, &rc] {
            rc[(size_t)k] = vo_pipeline_sequence(ctxs[k], imgs + (size_t)b * image_stride, e - b + 1, rows, cols, step,
                                                 image_stride, max_features, ratio, K, threshold, confidence, fixed_iters,
                                                 results + b);
        });
    }
    for (auto& w : workers) w.join();
    int out = VO_OK;
    for (int k = 0; k < n_ctx; ++k) {
        if (rc[(size_t)k] == VO_OK) continue;
        if (out == VO_OK || out == VO_ERR_CAPACITY) out = rc[(size_t)k];   // a hard failure outranks a truncation report
        if (k != 0) ctxs[0]->err = "context " + std::to_string(k) + ": " + ctxs[k]->err;
    }
    return out;
}
