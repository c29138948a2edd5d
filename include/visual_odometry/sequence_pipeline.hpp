// Batched form of the reference's per-pair loop (ref:src/main.cpp:152-184: match_images then
// estimate_motion for frames i, i+1): a run of consecutive frames goes to the GPU in one
// call, every frame is detected once, and one record per consecutive pair comes back.  The
// records equal what match_images_orb + estimate_motion return for each pair.
#pragma once

#include <span>
#include <vector>

#include <visual_odometry/compat/cv_min.hpp>
#include <visual_odometry/motion_estimator.hpp>

namespace visual_odometry {

struct sequence_step {
    int keypoints1{0};   // keypoints of frame i
    int keypoints2{0};   // keypoints of frame i + 1
    int matches{0};      // ratio-test survivors (match_result::matches.size())
    motion_estimate motion{};
};

// frames: at least two CV_8UC1 images of one size.  Returns frames.size() - 1 steps (an empty
// vector on bad input or a device failure - see visual_odometry::last_error()).
[[nodiscard]] auto process_sequence(std::span<cv::Mat const> frames, camera_intrinsics const& intrinsics,
                                    int max_features = 2000, float ratio_threshold = 0.75f,
                                    motion_estimator_config const& config = {}) -> std::vector<sequence_step>;

// The same over several GPUs of one node: `devices` lists the GPU indices; the pairs are split
// into one contiguous block per device (the frame at a block boundary is detected by both
// neighbours), every block runs on a host thread and context of its own, and the steps come
// back in pair order - the same steps as the single-GPU call.  An empty list means the
// calling thread's default device.  Calls from several host threads are serialised.
[[nodiscard]] auto process_sequence(std::span<cv::Mat const> frames, camera_intrinsics const& intrinsics,
                                    std::span<int const> devices, int max_features = 2000,
                                    float ratio_threshold = 0.75f, motion_estimator_config const& config = {})
    -> std::vector<sequence_step>;

}  // namespace visual_odometry
