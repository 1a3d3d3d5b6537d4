// fuzz_parsers.cpp — the two host parsers that read untrusted files, under AddressSanitizer + UBSan (make -C tests/host asan):
//   TGAImage::decode_tga (tinyrenderder_amd/shim/trgl_image.h; the reference's read_tga_file + load_rle_data, tgaimage.cpp:76-160,
//   which itself writes one pixel out of bounds on an overrunning RLE packet)  and  trgl_obj::load (shim/trgl_obj.h).
// Seeds: every file in tests/golden/tga_read/ and a few hand-written OBJ texts; mutations: truncation at every length, random
// byte flips / insertions / header rewrites (SplitMix64, fixed seed).  Any out-of-bounds access, overflow or leak aborts the run.
//   fuzz_parsers <tga_seed_dir> <scratch_dir> [iterations]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dirent.h>
#include <fstream>
#include <string>
#include <vector>

#include "../../tinyrenderder_amd/shim/trgl_image.h"
#include "../../tinyrenderder_amd/shim/trgl_obj.h"

static uint64_t rng_state = 0x5EED0F22ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static std::vector<uint8_t> slurp(const std::string& p) {
    std::ifstream in(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
}

static unsigned long long decoded = 0, rejected = 0;
static void try_tga(const std::vector<uint8_t>& f) {
    // keep allocations sane: the decoder allocates width*height*bpp up front, as the reference does
    if (f.size() >= 18) { unsigned w = f[12] | (f[13] << 8), h = f[14] | (f[15] << 8); if ((unsigned long long)w * h > (1ull << 22)) return; }
    TGAImage img;
    if (img.decode_tga(f.data(), f.size())) {
        ++decoded;
        // touch every byte the image claims to own
        volatile unsigned sum = 0;
        const uint8_t* b = img.buffer();
        for (size_t i = 0; i < size_t(img.width()) * img.height() * img.bytespp(); ++i) sum += b[i];
        (void)sum;
        (void)img.encode_tga(true, true);
    } else ++rejected;
}

static void mutate_tga(const std::vector<uint8_t>& seed, int iterations) {
    for (size_t len = 0; len <= seed.size(); ++len) try_tga(std::vector<uint8_t>(seed.begin(), seed.begin() + len));   // every truncation
    for (int it = 0; it < iterations; ++it) {
        std::vector<uint8_t> f = seed;
        const int edits = 1 + int(rnd() % 4);
        for (int e = 0; e < edits && !f.empty(); ++e) {
            const uint64_t r = rnd();
            const size_t pos = size_t(r >> 8) % f.size();
            switch (r & 7) {
            case 0: f[pos] ^= uint8_t(1u << ((r >> 40) & 7)); break;
            case 1: f[pos] = uint8_t(r >> 48); break;
            case 2: f.insert(f.begin() + pos, uint8_t(r >> 48)); break;
            case 3: f.erase(f.begin() + pos); break;
            case 4: if (f.size() >= 18) { f[12] = uint8_t(r >> 16) & 63; f[13] = 0; f[14] = uint8_t(r >> 24) & 63; f[15] = 0; } break;   // small random size
            case 5: if (f.size() >= 18) f[2] = uint8_t((r >> 16) % 13); break;                       // data type
            case 6: if (f.size() >= 18) f[16] = uint8_t(((r >> 16) % 6) * 8); break;                 // bits per pixel
            default: if (f.size() >= 18) { f[0] = uint8_t(r >> 16); f[17] = uint8_t(r >> 24); } break;   // id length, descriptor
            }
        }
        try_tga(f);
    }
}

static unsigned long long obj_ok = 0, obj_bad = 0;
static void try_obj(const std::string& text, const std::string& scratch) {
    const std::string p = scratch + "/fuzz.obj";
    { std::ofstream o(p, std::ios::binary); o << text; }
    trgl_obj::Mesh m;
    if (trgl_obj::load(p, m)) {
        ++obj_ok;
        const size_t nv = m.vertices.size() / 14;
        for (uint32_t i : m.indices) if (i >= nv) { std::fprintf(stderr, "index %u out of %zu vertices\n", i, nv); std::abort(); }
        if (m.indices.size() % 3) { std::fprintf(stderr, "indices not a multiple of 3\n"); std::abort(); }
    } else ++obj_bad;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: fuzz_parsers <tga_seed_dir> <scratch_dir> [iterations]\n"); return 2; }
    const int iterations = argc > 3 ? std::atoi(argv[3]) : 400;
    int nseeds = 0;
    if (DIR* d = opendir(argv[1])) {
        while (dirent* e = readdir(d)) {
            const std::string n = e->d_name;
            if (n.size() > 4 && n.substr(n.size() - 4) == ".tga") { mutate_tga(slurp(std::string(argv[1]) + "/" + n), iterations); ++nseeds; }
        }
        closedir(d);
    }
    if (!nseeds) { std::fprintf(stderr, "no .tga seeds in %s\n", argv[1]); return 2; }

    const char* obj_seeds[] = {
        "v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nvn 0 0 1\nf 1/1/1 2/2/1 3/3/1\n",
        "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\nf -4 -3 -2 -1\n",
        "# comment\nv 1e308 -1e308 nan\nv inf 2 3\nv 0 0\nf 1//1 2//2 3//3\nf 1/ 2/ 3/\n",
        "v 0 0 0\nf 1 1 1\nf 1\nf\nf 0 0 0\nf 99999999999999999999 1 1\nf -9223372036854775808 1 1\nf 1/-9223372036854775808/1 1 1\n",
        "vt\nvn\nv\nf 1/1/1 2/2/2 3/3/3\n",
    };
    const char alphabet[] = "vtnf 0123456789-+./e\n\r\t#xX";
    for (const char* seed : obj_seeds) {
        try_obj(seed, argv[2]);
        const std::string s0 = seed;
        for (size_t len = 0; len <= s0.size(); ++len) try_obj(s0.substr(0, len), argv[2]);
        for (int it = 0; it < iterations; ++it) {
            std::string s = s0;
            const int edits = 1 + int(rnd() % 6);
            for (int e = 0; e < edits && !s.empty(); ++e) {
                const uint64_t r = rnd();
                const size_t pos = size_t(r >> 8) % s.size();
                const char c = alphabet[(r >> 40) % (sizeof(alphabet) - 1)];
                if (r & 1) s[pos] = c; else if (r & 2) s.insert(s.begin() + pos, c); else s.erase(s.begin() + pos);
            }
            try_obj(s, argv[2]);
        }
    }
    std::printf("tga: %d seeds, %llu decoded, %llu rejected; obj: %llu loaded, %llu rejected; no sanitizer report\n",
                nseeds, decoded, rejected, obj_ok, obj_bad);
    return 0;
}
