// Fused frame-pair pipeline: detect x2 -> kNN(2) + ratio -> E-RANSAC + pose,
// every intermediate resident in HBM.  Stands in for match_images_orb
// (ref:src/image_matcher.cpp:71-81) followed by estimate_motion as the
// reference's main loop chains them (ref:src/main.cpp:162-170).
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

}  // namespace vo

using namespace vo;

// Shared body: n_images frames, pair p = (p * pair_step, p * pair_step + 1).
static int pipeline_run(vo_ctx* c, const uint8_t* d_imgs, int n_images, int n_pairs, int pair_step, int rows, int cols,
                        int step, size_t image_stride, int max_features, float ratio, const double K[4],
                        double threshold, double confidence, int fixed_iters, vo_pair_result* d_results) {
    VO_HIP(c, hipSetDevice(c->device));
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
    const size_t dstride = (size_t)pair_step * cap * 32;
    if ((rc = knn2_launch(c, desc, cnt, pair_step, cap, dstride, desc + (size_t)cap * 32, cnt + 1, cap, dstride, n_pairs,
                          (int32_t*)c->st_idx.p, (int32_t*)c->st_dist.p, max_features)))
        return rc;
    // Everything after the kNN search - ratio test, counts, RANSAC rounds, pose recovery - is a
    // chain of small, latency-bound kernels: it goes to the side stream, so that the main stream
    // can start the next batch's detection right behind the matcher.  ev_tail marks the point
    // after which the side stream no longer reads this batch's keypoints and kNN table.
    hipStream_t side = c->side;
    VO_HIP(c, hipEventRecord(c->ev_init, c->stream));
    VO_HIP(c, hipStreamWaitEvent(side, c->ev_init, 0));
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
    VO_HIP(c, hipGetLastError());
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
