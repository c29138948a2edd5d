// Mutation fuzzer for the image decoders (PNG, PGM/PPM, sequential JPEG), meant to be built with
// -fsanitize=address,undefined: fuzz_decoders <iterations> <seed> <file>...
// Every seed file is decoded as it is and then `iterations` times with random damage (flipped
// bytes, overwritten runs, truncation, a spliced tail of another file).  A decoder may refuse a
// file; it may not read or write out of bounds, overflow or hang.  Prints the number of files
// that still decoded.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../visual_odometry_amd/csrc/host/vo_image_loader.cpp"   // the decoders live in its unnamed namespace

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    long const iterations = std::atol(argv[1]);
    std::mt19937_64 rng((unsigned long long)std::atoll(argv[2]));
    std::vector<std::vector<uint8_t>> seeds;
    for (int i = 3; i < argc; ++i) {
        std::vector<uint8_t> f;
        if (!visual_odometry::read_file(argv[i], f)) {
            std::fprintf(stderr, "cannot read %s\n", argv[i]);
            return 2;
        }
        seeds.push_back(std::move(f));
    }
    auto decode = [](std::vector<uint8_t> const& f) -> bool {
        cv::Mat m;
        if (f.size() >= 2 && f[0] == 0xFF && f[1] == 0xD8) m = visual_odometry::detail::decode_jpeg_gray(f);
        else if (f.size() >= 2 && f[0] == 'P') m = visual_odometry::decode_pnm(f);
        else m = visual_odometry::decode_png(f);
        if (m.empty()) return false;
        unsigned long sum = 0;   // touch every pixel: the buffer is as large as the header says
        for (int y = 0; y < m.rows; ++y)
            for (int x = 0; x < m.cols; ++x) sum += m.ptr<uint8_t>(y)[x];
        return sum != ~0ul;
    };
    long ok = 0;
    for (auto const& s : seeds) ok += decode(s);
    long const whole = ok;
    for (long it = 0; it < iterations; ++it) {
        std::vector<uint8_t> f = seeds[rng() % seeds.size()];
        int const kind = (int)(rng() % 5);
        if (kind == 0) {   // a few flipped bytes anywhere
            for (int k = 0, n = 1 + (int)(rng() % 6); k < n; ++k) f[rng() % f.size()] ^= (uint8_t)(1u << (rng() % 8));
        } else if (kind == 1) {   // damage in the headers
            for (int k = 0, n = 1 + (int)(rng() % 4); k < n; ++k) f[rng() % std::min<size_t>(f.size(), 96)] = (uint8_t)rng();
        } else if (kind == 2) {   // truncation
            f.resize(1 + rng() % f.size());
        } else if (kind == 3) {   // a run of one value
            size_t const a = rng() % f.size(), n = std::min<size_t>(f.size() - a, 1 + rng() % 64);
            std::fill(f.begin() + (long)a, f.begin() + (long)(a + n), (uint8_t)(rng() % 3 == 0 ? 0xFF : rng()));
        } else {   // the tail of another file
            auto const& o = seeds[rng() % seeds.size()];
            size_t const cut = rng() % f.size(), from = rng() % o.size();
            f.resize(cut);
            f.insert(f.end(), o.begin() + (long)from, o.end());
        }
        ok += decode(f);
    }
    std::printf("%ld of %zu seed files and %ld of %ld damaged files decoded\n", whole, seeds.size(), ok - whole, iterations);
    return whole == (long)seeds.size() ? 0 : 1;
}
