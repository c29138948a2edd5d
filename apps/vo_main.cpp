// Command-line front end with the reference tool's flags and progress format
// (ref:src/main.cpp:30-199): walks an image directory, estimates the motion between
// consecutive frames and writes the trajectory JSON.  Frames go to the GPU a batch at a time
// (process_sequence), so every frame is decoded, uploaded and detected once.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <future>
#include <iomanip>
#include <iostream>
#include <optional>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include <visual_odometry/image_loader.hpp>
#include <visual_odometry/image_matcher.hpp>
#include <visual_odometry/motion_estimator.hpp>
#include <visual_odometry/sequence_pipeline.hpp>
#include <visual_odometry/trajectory.hpp>
#include <vo_hip.h>

namespace {

struct options {
    std::string image_dir;
    std::string camera_yaml{"data/kitti_camera.yaml"};
    std::string output_path{"trajectory.json"};
    std::string matcher{"orb"};
    int max_frames{0};
    int batch{64};   // frame pairs per GPU call
    int gpus{1};     // GPUs the pairs of a batch are split over (devices 0 .. gpus - 1)
    int decode_threads{(int)std::clamp(std::thread::hardware_concurrency(), 1u, 16u)};   // host threads that decode a batch's files
};

void usage(std::string_view program) {
    std::cerr << "Usage: " << program << " --images <dir> [options]\n"
              << "\nOptions:\n"
              << "  --images <dir>     Path to image directory\n"
              << "  --camera <file>    Camera intrinsics YAML (default: data/kitti_camera.yaml)\n"
              << "  --output <file>    Output trajectory JSON (default: trajectory.json)\n"
              << "  --matcher <name>   Feature matcher: orb (default)\n"
              << "  --max-frames <n>   Maximum frames to process (default: all)\n"
              << "  --batch <n>        Frame pairs per GPU call (default: 64)\n"
              << "  --gpus <n>         GPUs to split every batch over, devices 0..n-1 (default: 1)\n"
              << "  --decode-threads <n>  Host threads that decode a batch's image files (default: the host's cores, at most 16)\n"
              << "  --help             Show this help message\n";
}

auto parse(int argc, char** argv) -> std::optional<options> {
    options o;
    for (int i = 1; i < argc; ++i) {
        std::string_view const a = argv[i];
        auto value = [&]() -> char const* { return i + 1 < argc ? argv[++i] : nullptr; };
        char const* v = nullptr;
        if (a == "--help" || a == "-h") {
            usage(argv[0]);
            return std::nullopt;
        } else if (a == "--images" && (v = value())) o.image_dir = v;
        else if (a == "--camera" && (v = value())) o.camera_yaml = v;
        else if (a == "--output" && (v = value())) o.output_path = v;
        else if (a == "--matcher" && (v = value())) o.matcher = v;
        else if (a == "--max-frames" && (v = value())) o.max_frames = std::atoi(v);
        else if (a == "--batch" && (v = value())) o.batch = std::max(1, std::atoi(v));
        else if (a == "--decode-threads" && (v = value())) o.decode_threads = std::max(1, std::atoi(v));
        else if (a == "--gpus" && (v = value())) o.gpus = std::max(1, std::atoi(v));
        else {
            std::cerr << "Unknown argument: " << a << "\n";
            usage(argv[0]);
            return std::nullopt;
        }
    }
    if (o.image_dir.empty()) {
        std::cerr << "Error: --images is required\n";
        usage(argv[0]);
        return std::nullopt;
    }
    return o;
}

}  // namespace

int main(int argc, char** argv) {
    namespace vo = visual_odometry;
    auto const opt = parse(argc, argv);
    if (!opt) return 1;

    std::cout << "Visual Odometry Pipeline\n"
              << "  Images:     " << opt->image_dir << "\n"
              << "  Camera:     " << opt->camera_yaml << "\n"
              << "  Output:     " << opt->output_path << "\n"
              << "  Matcher:    " << opt->matcher << "\n"
              << "  Max frames: " << (opt->max_frames > 0 ? std::to_string(opt->max_frames) : "all") << "\n\n";

    auto const intrinsics = vo::camera_intrinsics::load_from_yaml(opt->camera_yaml);
    if (!intrinsics.has_value()) {
        std::cerr << "Error: " << intrinsics.error() << "\n";
        return 1;
    }
    std::cout << "Loaded camera intrinsics (fx=" << intrinsics->fx << ", fy=" << intrinsics->fy << ")\n";

    auto loader = vo::image_loader::create(opt->image_dir);
    if (!loader.has_value()) {
        std::cerr << "Error: " << loader.error() << "\n";
        return 1;
    }
    std::cout << "Found " << loader->size() << " images\n";
    if (loader->has_timestamps()) std::cout << "Timestamps loaded from rgb.txt\n";
    if (loader->size() < 2) {
        std::cerr << "Error: Need at least 2 images for visual odometry\n";
        return 1;
    }

    int max_features = 2000;
    float ratio = 0.75f;
    try {
        auto const matcher = vo::create_image_matcher(opt->matcher);   // validates the name as the reference does
        std::cout << "Using matcher: " << std::visit([](auto const& m) { return std::string{m.name()}; }, matcher) << "\n";
    } catch (std::exception const& e) {
        std::cerr << "Error: " << e.what() << "\n";
        return 1;
    }

    vo::Trajectory trajectory(loader->get_timestamp(0));
    size_t const total_pairs = loader->size() - 1;
    size_t const max_pairs = opt->max_frames > 0 ? std::min((size_t)opt->max_frames, total_pairs) : total_pairs;
    std::cout << "\nProcessing " << max_pairs << " frame pairs...\n";

    std::vector<int> devices;
    if (opt->gpus > 1) {
        int const have = vo_device_count();
        if (opt->gpus > have) {
            std::cerr << "Error: --gpus " << opt->gpus << " but this machine has " << have << " GPU(s)\n";
            return 1;
        }
        for (int d = 0; d < opt->gpus; ++d) devices.push_back(d);
    }

    int valid_motions = 0;
    // one pair's outcome, reported and accumulated in pair order
    auto account = [&](size_t i, vo::sequence_step const& st) {
        if (trajectory.add_motion(st.motion, loader->get_timestamp(i + 1))) ++valid_motions;
        if ((i + 1) % 100 == 0 || i == max_pairs - 1)
            std::cout << "  Frame " << std::setw(5) << (i + 1) << "/" << max_pairs << " | Matches: " << st.matches
                      << " | Inliers: " << st.motion.inliers << (st.motion.valid ? "" : " [FAILED]") << "\n";
    };
    // a pair on its own, as the reference's loop treats every pair (ref:src/main.cpp:162-170): the
    // route for pairs that cannot ride in a batch (frames of different size, a batch that failed)
    auto single_pair = [&](cv::Mat const& a, cv::Mat const& b) {
        vo::sequence_step st;
        auto const m = vo::match_images_orb(a, b, vo::feature_detector_config{.max_features = max_features},
                                            vo::feature_matcher_config{.ratio_threshold = ratio});
        st.matches = (int)m.matches.size();
        st.motion = vo::estimate_motion(m.points1, m.points2, *intrinsics);
        return st;
    };
    // Decoding is what the host spends its time on (milliseconds per frame against microseconds
    // on the GPU), and frames decode independently: a batch's files are shared out over the host's
    // cores, each worker taking the next index, and the next batch is decoded while the GPU works
    // on this one.
    struct decoded_batch {
        std::vector<std::optional<cv::Mat>> frames;   // frames first .. first + n_pairs (nullopt: failed to load)
        std::vector<std::string> load_error;
    };
    auto decode_batch = [&](size_t first, size_t n_pairs, bool skip_first) {
        decoded_batch d{std::vector<std::optional<cv::Mat>>(n_pairs + 1), std::vector<std::string>(n_pairs + 1)};
        std::atomic<size_t> next{skip_first ? size_t{1} : size_t{0}};   // frame 0 is the previous batch's last one
        auto decode = [&] {
            for (size_t k = next.fetch_add(1); k <= n_pairs; k = next.fetch_add(1)) {
                try {   // (an exception that left a worker thread would end the process)
                    auto img = loader->load_image(first + k);
                    if (img.has_value() && !img.value().empty()) d.frames[k] = img.value();
                    else d.load_error[k] = img.has_value() ? std::string("Failed to load image") : img.error();
                } catch (std::exception const& e) {
                    d.frames[k].reset();
                    d.load_error[k] = std::string("Failed to load image: ") + e.what();
                }
            }
        };
        size_t const workers = std::min<size_t>({(size_t)opt->decode_threads, n_pairs + 1});
        std::vector<std::thread> pool;
        for (size_t w = 1; w < workers; ++w) pool.emplace_back(decode);
        decode();
        for (auto& t : pool) t.join();
        return d;
    };
    auto pairs_from = [&](size_t first) { return std::min((size_t)opt->batch, max_pairs - first); };
    std::future<decoded_batch> ahead;
    if (max_pairs > 0) ahead = std::async(std::launch::async, decode_batch, size_t{0}, pairs_from(0), false);
    std::optional<cv::Mat> carried;   // last frame of the previous batch (nullopt: it failed to load)
    std::string carried_error;
    for (size_t first = 0; first < max_pairs; first += (size_t)opt->batch) {
        size_t const n_pairs = pairs_from(first);
        // frames first .. first + n_pairs; a frame that fails to load costs only the pairs it
        // belongs to, as in the reference (ref:src/main.cpp:152-158)
        decoded_batch cur = ahead.get();
        std::vector<std::optional<cv::Mat>>& frames = cur.frames;
        std::vector<std::string>& load_error = cur.load_error;
        if (first > 0) {
            frames[0] = carried;
            load_error[0] = carried_error;
        }
        if (size_t const next_first = first + (size_t)opt->batch; next_first < max_pairs)
            ahead = std::async(std::launch::async, decode_batch, next_first, pairs_from(next_first), true);
        std::vector<std::optional<vo::sequence_step>> steps(n_pairs);
        // maximal runs of loaded frames of one size and type go to the GPU in one call each
        size_t k = 0;
        while (k <= n_pairs) {
            if (!frames[k]) { ++k; continue; }
            size_t e = k + 1;
            while (e <= n_pairs && frames[e] && frames[e]->rows == frames[k]->rows && frames[e]->cols == frames[k]->cols &&
                   frames[e]->type() == frames[k]->type())
                ++e;
            if (e - k >= 2 && frames[k]->type() == CV_8UC1) {
                std::vector<cv::Mat> run;
                for (size_t j = k; j < e; ++j) run.push_back(*frames[j]);
                auto const got = vo::process_sequence(run, *intrinsics, devices, max_features, ratio);
                if (got.size() == run.size() - 1) {
                    for (size_t j = 0; j < got.size(); ++j) steps[k + j] = got[j];
                } else {
                    std::cerr << "Warning: batch of frames " << (first + k) << ".." << (first + e - 1)
                              << " failed on the GPU (" << vo::last_error() << "); processing its pairs one by one\n";
                }
            }
            k = e;
        }
        for (size_t j = 0; j < n_pairs; ++j) {
            size_t const i = first + j;
            if (!frames[j] || !frames[j + 1]) {
                std::cerr << "Warning: Failed to load pair " << i << ": "
                          << (!frames[j] ? load_error[j] : load_error[j + 1]) << "\n";
                continue;
            }
            if (!steps[j]) steps[j] = single_pair(*frames[j], *frames[j + 1]);
            account(i, *steps[j]);
        }
        carried = frames[n_pairs];
        carried_error = load_error[n_pairs];
    }

    std::cout << "\nProcessed " << max_pairs << " pairs, " << valid_motions << " valid motions\n";
    auto const saved = trajectory.save_to_json(opt->output_path);
    if (!saved) {
        std::cerr << "Error: Failed to save trajectory: " << saved.error() << "\n";
        return 1;
    }
    std::cout << "Saved trajectory with " << trajectory.size() << " poses to: " << opt->output_path << "\n";
    return 0;
}
