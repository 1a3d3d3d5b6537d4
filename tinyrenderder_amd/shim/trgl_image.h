// trgl_image.h — TGAColor / TGAImage with the reference's interface (tgaimage.h:29-104) and the
// byte-exact TGA file format of its writer (tgaimage.cpp:161-242): 18-byte header, no footer,
// imagedescriptor 0x00 when vflip (the default), RLE packets formed exactly as the reference forms
// them (its raw packets run up to AND INCLUDING the first pixel of the next repeated pair).
// The reader follows read_tga_file / load_rle_data (tgaimage.cpp:76-160).  Host-only; this is SURVEY.md §8(f) row N3.
#pragma once
#include <cstdint>
#include <cstring>
#include <fstream>
#include <iterator>
#include <string>
#include <vector>

struct TGAColor {
    std::uint8_t bgra[4] = { 0, 0, 0, 255 };
    std::uint8_t bytespp = 4;
    TGAColor() = default;
    TGAColor(std::uint8_t R, std::uint8_t G, std::uint8_t B, std::uint8_t A = 255) : bgra{ B, G, R, A }, bytespp(4) {}
    TGAColor(std::uint8_t v) : bgra{ v, v, v, 255 }, bytespp(1) {}
    TGAColor(const std::uint8_t* p, std::uint8_t bpp) : bgra{ 0, 0, 0, 0 }, bytespp(bpp) { for (int i = 0; i < bpp; ++i) bgra[i] = p[i]; }
    std::uint8_t& operator[](int i) { return bgra[i]; }
    const std::uint8_t& operator[](int i) const { return bgra[i]; }
    TGAColor operator*(float k) const {
        TGAColor r = *this;
        if (k < 0.f) k = 0.f;
        if (k > 1.f) k = 1.f;
        for (int i = 0; i < 4; ++i) r.bgra[i] = std::uint8_t(bgra[i] * k);
        return r;
    }
    std::uint32_t packed() const { return bgra[0] | (std::uint32_t(bgra[1]) << 8) | (std::uint32_t(bgra[2]) << 16) | (std::uint32_t(bgra[3]) << 24); }
};

class TGAImage {
public:
    enum Format { GRAYSCALE = 1, RGB = 3, RGBA = 4 };
    TGAImage() = default;
    TGAImage(int width, int height, int bytespp, TGAColor clear = TGAColor()) : w(width), h(height), bpp(std::uint8_t(bytespp)) {
        data.resize(std::size_t(w) * h * bpp);
        for (std::size_t i = 0; i < std::size_t(w) * h; ++i) for (int k = 0; k < bpp; ++k) data[i * bpp + k] = clear.bgra[k];
    }
    int width() const { return w; }
    int height() const { return h; }
    int bytespp() const { return bpp; }
    std::uint8_t* buffer() { return data.empty() ? nullptr : data.data(); }
    const std::uint8_t* buffer() const { return data.empty() ? nullptr : data.data(); }
    TGAColor get(int x, int y) const {
        if (data.empty() || x < 0 || y < 0 || x >= w || y >= h) return TGAColor();
        return TGAColor(&data[(std::size_t(x) + std::size_t(y) * w) * bpp], bpp);
    }
    void set(int x, int y, const TGAColor& c) {
        if (data.empty() || x < 0 || y < 0 || x >= w || y >= h) return;
        std::memcpy(&data[(std::size_t(x) + std::size_t(y) * w) * bpp], c.bgra, bpp);
    }
    void flip_vertically() {
        std::size_t line = std::size_t(w) * bpp;
        std::vector<std::uint8_t> tmp(line);
        for (int y = 0; y < h / 2; ++y) {
            std::uint8_t* a = &data[y * line]; std::uint8_t* b = &data[(h - 1 - y) * line];
            std::memcpy(tmp.data(), a, line); std::memcpy(a, b, line); std::memcpy(b, tmp.data(), line);
        }
    }
    void flip_horizontally() {
        for (int y = 0; y < h; ++y) for (int x = 0; x < w / 2; ++x) for (int k = 0; k < bpp; ++k)
            std::swap(data[(std::size_t(x) + std::size_t(y) * w) * bpp + k], data[(std::size_t(w - 1 - x) + std::size_t(y) * w) * bpp + k]);
    }

    // The exact bytes TGAImage::write_tga_file(name, vflip, rle) puts on disk.
    std::vector<std::uint8_t> encode_tga(bool vflip = true, bool rle = true) const {
        std::vector<std::uint8_t> out(18, 0);
        out[2] = bpp == 1 ? (rle ? 11 : 3) : (rle ? 10 : 2);
        out[12] = std::uint8_t(w & 0xff); out[13] = std::uint8_t((w >> 8) & 0xff);
        out[14] = std::uint8_t(h & 0xff); out[15] = std::uint8_t((h >> 8) & 0xff);
        out[16] = std::uint8_t(bpp * 8);
        out[17] = vflip ? 0x00 : 0x20;
        if (!rle) { out.insert(out.end(), data.begin(), data.end()); return out; }
        const int npix = w * h, maxrun = 128;
        auto same = [&](int a, int b) { return std::memcmp(&data[std::size_t(a) * bpp], &data[std::size_t(b) * bpp], bpp) == 0; };
        for (int cur = 0; cur < npix;) {
            int run = 1;
            while (cur + run < npix && run < maxrun && same(cur + run, cur)) ++run;
            if (run > 1) {                                   // repeat packet
                out.push_back(std::uint8_t(run - 1 + 128));
                out.insert(out.end(), &data[std::size_t(cur) * bpp], &data[std::size_t(cur) * bpp] + bpp);
            } else {                                         // literal packet: stops AFTER meeting an equal neighbour pair
                while (cur + run < npix && run < maxrun && !same(cur + run, cur + run - 1)) ++run;
                out.push_back(std::uint8_t(run - 1));
                out.insert(out.end(), &data[std::size_t(cur) * bpp], &data[std::size_t(cur) * bpp] + std::size_t(run) * bpp);
            }
            cur += run;
        }
        return out;
    }
    bool write_tga_file(const std::string& filename, bool vflip = true, bool rle = true) const {
        std::ofstream out(filename, std::ios::binary);
        if (!out.is_open()) return false;
        auto bytes = encode_tga(vflip, rle);
        out.write(reinterpret_cast<const char*>(bytes.data()), std::streamsize(bytes.size()));
        return bool(out);
    }
    // TGAImage::read_tga_file (tgaimage.cpp:76-126) + load_rle_data (:128-160) on a file image in memory, with the
    // reference's std::ifstream behaviour spelled out: a short read delivers what is there and fails the stream, after
    // which every read is a no-op - so a truncated raw image keeps zeros, a truncated RLE stream repeats the last
    // colour read (chunk headers then read as 0 = one raw pixel), and neither is an error.  Errors (false) are: header
    // shorter than 18 bytes, width/height 0, bits per pixel not 8/24/32, data type not 2/3/10/11, and an RLE packet
    // running past the last pixel (the reference writes one pixel out of bounds before it notices; that write is not
    // reproduced).
    bool decode_tga(const std::uint8_t* file, std::size_t size) {
        data.clear();
        std::size_t pos = 0; bool good = true;
        auto rd = [&](std::uint8_t* dst, std::size_t k) {
            if (!good) return;
            std::size_t avail = pos < size ? size - pos : 0, m = k < avail ? k : avail;
            if (m) std::memcpy(dst, file + pos, m);
            pos += m;
            if (m < k) good = false;
        };
        std::uint8_t hd[18] = { 0 };
        rd(hd, 18);
        if (!good) return false;                                             // :86-90
        w = hd[12] | (hd[13] << 8); h = hd[14] | (hd[15] << 8); bpp = std::uint8_t(hd[16] >> 3);   // :92-94
        if (w <= 0 || h <= 0 || (bpp != 1 && bpp != 3 && bpp != 4)) return false;                // :96-99
        data.assign(std::size_t(w) * h * bpp, 0);                            // :101
        pos += hd[0];                                                        // :103 seekg(idlength, cur)
        if (hd[2] == 2 || hd[2] == 3) {
            rd(data.data(), data.size());                                    // :105-108
        } else if (hd[2] == 10 || hd[2] == 11) {
            const int npix = w * h;
            int cur = 0;
            TGAColor c;                                                      // :132, {0,0,0,255}
            while (cur < npix) {
                std::uint8_t head = 0;
                rd(&head, 1);
                const bool raw = head < 128;
                const int count = raw ? head + 1 : head - 127;
                if (!raw) rd(c.bgra, bpp);
                for (int i = 0; i < count; ++i) {
                    if (raw) rd(c.bgra, bpp);
                    if (cur >= npix) return false;                           // :145 / :155
                    std::memcpy(&data[std::size_t(cur++) * bpp], c.bgra, bpp);
                }
            }
        } else return false;                                                 // :113-116
        if (!(hd[17] & 0x20)) flip_vertically();                             // :118
        if (hd[17] & 0x10) flip_horizontally();                              // :119
        return true;
    }
    bool read_tga_file(const std::string& filename) {
        std::ifstream in(filename, std::ios::binary);
        if (!in.is_open()) return false;
        std::vector<std::uint8_t> bytes((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        return decode_tga(bytes.data(), bytes.size());
    }

private:
    int w = 0, h = 0;
    std::uint8_t bpp = 0;
    std::vector<std::uint8_t> data;
};
