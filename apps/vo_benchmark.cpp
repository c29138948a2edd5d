// Matcher benchmark with the reference tool's flags, per-pair lines and JSON keys
// (ref:benchmarks/benchmark_matchers.cpp:25-192): wall time of match_images per consecutive
// pair (motion estimation outside the timed region), average matches / inliers / inlier ratio.
// On top of the per-call numbers it reports the batched GPU form (all frames in one
// process_sequence call, motion estimation included) under "batched_*" keys.
#include <chrono>
#include <cstdio>
#include <fstream>
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

namespace vo = visual_odometry;

namespace {

struct options {
    std::string image_dir, matcher{"orb"}, output_path{"benchmark_results.json"}, camera_yaml;
};

void usage(std::string_view program) {
    std::cerr << "Usage: " << program << " --images <dir> [options]\n"
              << "\nOptions:\n"
              << "  --images <dir>     Path to image directory\n"
              << "  --matcher <name>   Feature matcher: orb (default)\n"
              << "  --output <file>    Output JSON file (default: benchmark_results.json)\n"
              << "  --camera <file>    Camera intrinsics YAML (for accuracy metrics)\n"
              << "  --help             Show this help message\n";
}

auto parse(int argc, char** argv) -> std::optional<options> {
    options o;
    for (int i = 1; i < argc; ++i) {
        std::string_view const a = argv[i];
        if (a == "--help" || a == "-h") {
            usage(argv[0]);
            return std::nullopt;
        }
        if (i + 1 < argc && a == "--images") o.image_dir = argv[++i];
        else if (i + 1 < argc && a == "--matcher") o.matcher = argv[++i];
        else if (i + 1 < argc && a == "--output") o.output_path = argv[++i];
        else if (i + 1 < argc && a == "--camera") o.camera_yaml = argv[++i];
        else {
            std::cerr << "Unknown option: " << a << '\n';
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

auto now_ms() -> double {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

int main(int argc, char** argv) {
    auto const opt = parse(argc, argv);
    if (!opt) return 1;
    auto loader = vo::image_loader::create(opt->image_dir);
    if (!loader.has_value()) {
        std::cerr << "Error: " << loader.error() << '\n';
        return 1;
    }
    std::cout << "Found " << loader->size() << " images\n";
    if (loader->size() < 2) {
        std::cerr << "Error: Need at least 2 images\n";
        return 1;
    }
    std::optional<vo::camera_intrinsics> intrinsics;
    if (!opt->camera_yaml.empty()) {
        auto k = vo::camera_intrinsics::load_from_yaml(opt->camera_yaml);
        if (!k.has_value()) {
            std::cerr << "Error: " << k.error() << '\n';
            return 1;
        }
        intrinsics = k.value();
    }
    std::string name;
    try {
        name = std::visit([](auto const& m) { return std::string{m.name()}; }, vo::create_image_matcher(opt->matcher));
    } catch (std::exception const& e) {
        std::cerr << "Error: " << e.what() << '\n';
        return 1;
    }
    vo::orb_matcher const matcher;
    std::cout << "Benchmarking " << name << "...\n";

    size_t pairs = 0, total_matches = 0, total_inliers = 0;
    double total_ms = 0;
    std::vector<cv::Mat> frames;
    loader->reset();
    while (loader->has_next()) {
        auto pair = loader->next_pair();
        if (!pair) {
            std::cerr << "Error loading pair: " << pair.error() << '\n';
            break;
        }
        if (frames.empty()) frames.push_back(pair->first);
        frames.push_back(pair->second);
        double const t0 = now_ms();
        auto const m = matcher.match_images(pair->first, pair->second);
        double const dt = now_ms() - t0;
        total_ms += dt;
        ++pairs;
        total_matches += m.points1.size();
        size_t inliers = 0;
        if (intrinsics && m.points1.size() >= 5) {
            inliers = (size_t)vo::estimate_motion(m.points1, m.points2, *intrinsics).inliers;
            total_inliers += inliers;
        }
        std::cout << "  Pair " << pairs << ": " << std::fixed << std::setprecision(2) << dt << " ms (" << m.points1.size()
                  << " matches";
        if (intrinsics) std::cout << ", " << inliers << " inliers";
        std::cout << ")\n";
    }

    // the same pairs in one batched call (needs intrinsics: it also estimates the motions)
    double batched_ms = 0;
    size_t batched_inliers = 0;
    bool same_size = true;
    for (auto const& f : frames) same_size = same_size && f.rows == frames[0].rows && f.cols == frames[0].cols;
    if (intrinsics && frames.size() >= 2 && same_size) {
        (void)vo::process_sequence(frames, *intrinsics);   // warm-up: allocations
        double const t0 = now_ms();
        auto const steps = vo::process_sequence(frames, *intrinsics);
        batched_ms = now_ms() - t0;
        for (auto const& s : steps) batched_inliers += (size_t)s.motion.inliers;
    }

    double const n = pairs ? (double)pairs : 1.0;
    std::ofstream file{opt->output_path};
    if (!file) {
        std::cerr << "Error: Failed to open output file: " << opt->output_path << '\n';
        return 1;
    }
    char buf[1024];
    std::snprintf(buf, sizeof buf,
                  "{\n  \"matcher\": \"%s\",\n  \"num_pairs\": %zu,\n  \"total_time_ms\": %.2f,\n  \"avg_time_ms\": %.2f,\n"
                  "  \"total_matches\": %zu,\n  \"avg_matches\": %.2f,\n  \"total_inliers\": %zu,\n  \"avg_inliers\": %.2f,\n"
                  "  \"avg_inlier_ratio\": %.4f,\n  \"batched_total_time_ms\": %.2f,\n  \"batched_avg_time_ms\": %.4f,\n"
                  "  \"batched_total_inliers\": %zu\n}\n",
                  name.c_str(), pairs, total_ms, pairs ? total_ms / n : 0.0, total_matches, pairs ? total_matches / n : 0.0,
                  total_inliers, (pairs && intrinsics) ? total_inliers / n : 0.0,
                  (intrinsics && total_matches) ? (double)total_inliers / (double)total_matches : 0.0, batched_ms,
                  pairs ? batched_ms / n : 0.0, batched_inliers);
    file << buf;
    std::cout << "\nResults written to: " << opt->output_path << '\n';
    return 0;
}
