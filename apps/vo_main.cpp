// Command-line front end with the reference tool's flags and progress format
// (ref:src/main.cpp:30-199): walks an image directory, estimates the motion between
// consecutive frames and writes the trajectory JSON.  Frames go to the GPU a batch at a time
// (process_sequence), so every frame is decoded, uploaded and detected once.
#include <algorithm>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <optional>
#include <string>
#include <string_view>
#include <vector>

#include <visual_odometry/image_loader.hpp>
#include <visual_odometry/image_matcher.hpp>
#include <visual_odometry/motion_estimator.hpp>
#include <visual_odometry/sequence_pipeline.hpp>
#include <visual_odometry/trajectory.hpp>

namespace {

struct options {
    std::string image_dir;
    std::string camera_yaml{"data/kitti_camera.yaml"};
    std::string output_path{"trajectory.json"};
    std::string matcher{"orb"};
    int max_frames{0};
    int batch{64};   // frame pairs per GPU call
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

    int valid_motions = 0;
    std::optional<cv::Mat> carried;   // last frame of the previous batch
    for (size_t first = 0; first < max_pairs; first += (size_t)opt->batch) {
        size_t const n_pairs = std::min((size_t)opt->batch, max_pairs - first);
        std::vector<cv::Mat> frames;
        std::vector<size_t> index;   // frame index of each loaded image
        bool failed = false;
        for (size_t k = 0; k <= n_pairs && !failed; ++k) {
            if (k == 0 && carried) {
                frames.push_back(*carried);
                index.push_back(first);
                continue;
            }
            auto img = loader->load_image(first + k);
            if (!img.has_value()) {
                std::cerr << "Warning: Failed to load pair " << (first + (k ? k - 1 : 0)) << ": " << img.error() << "\n";
                failed = true;
                break;
            }
            frames.push_back(img.value());
            index.push_back(first + k);
        }
        carried.reset();
        if (failed || frames.size() < 2) continue;
        auto const steps = vo::process_sequence(frames, *intrinsics, max_features, ratio);
        if (steps.size() != frames.size() - 1) {
            std::cerr << "Warning: batch starting at frame " << first << " failed: " << vo::last_error() << "\n";
            continue;
        }
        for (size_t k = 0; k < steps.size(); ++k) {
            size_t const i = first + k;
            if (trajectory.add_motion(steps[k].motion, loader->get_timestamp(i + 1))) ++valid_motions;
            if ((i + 1) % 100 == 0 || i == max_pairs - 1)
                std::cout << "  Frame " << std::setw(5) << (i + 1) << "/" << max_pairs << " | Matches: " << steps[k].matches
                          << " | Inliers: " << steps[k].motion.inliers << (steps[k].motion.valid ? "" : " [FAILED]") << "\n";
        }
        carried = frames.back();
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
