// demo_multi.cpp - the multi-GPU frame from a C++ host: ONE PROCESS PER GPU over the C ABI of include/trgl.h.
//
//   demo_multi <scene.bin> <out.bin> <rank> <world> <id-file> [bands]
//
// Every rank submits the whole triangle stream (setup is replicated: 96 B per triangle of HBM reads are cheaper than moving
// 128-byte records over xGMI), owns the screen rows trgl_set_strip (or, with `bands`, trgl_set_interleave) gives it, and
// trgl_gather joins the rows of all ranks in place with RCCL all-gathers - the reference's final TGAImage, complete on every GPU.
// The RCCL bootstrap needs no RCCL headers here: rank 0 writes trgl_rccl_unique_id() to <id-file>, the others wait for it.
// scene.bin: int32 W, H, bpp, n; n x 12 doubles (clip coordinates); n x uint32 (colours).
// out.bin (every rank writes <out.bin>.<rank>): framebuffer bytes, depths, the print_render_stats() line of THIS rank.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../include/trgl.h"

#define CHK(call) do { int rc_ = (call); if (rc_ != TRGL_OK) { std::fprintf(stderr, "rank %d: %s failed (%d): %s\n", rank, #call, rc_, trgl_last_error(ctx)); return 3; } } while (0)

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: demo_multi <scene.bin> <out.bin> <rank> <world> <id-file> [bands]\n"); return 1; }
    const int rank = std::atoi(argv[3]), world = std::atoi(argv[4]);
    const bool bands = argc > 6 && std::strcmp(argv[6], "bands") == 0;
    trgl_ctx* ctx = nullptr;
    std::ifstream in(argv[1], std::ios::binary);
    int32_t hd[4];
    in.read(reinterpret_cast<char*>(hd), sizeof hd);
    const int W = hd[0], H = hd[1], bpp = hd[2]; const size_t n = size_t(hd[3]);
    std::vector<double> clip(n * 12); std::vector<uint32_t> colors(n);
    in.read(reinterpret_cast<char*>(clip.data()), std::streamsize(clip.size() * 8));
    in.read(reinterpret_cast<char*>(colors.data()), std::streamsize(colors.size() * 4));
    if (!in) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }

    // ---- RCCL bootstrap: the id travels through a file
    uint8_t id[TRGL_RCCL_ID_BYTES];
    const std::string id_path = argv[5];
    if (rank == 0) {
        CHK(trgl_rccl_unique_id(id));
        { std::ofstream f(id_path + ".tmp", std::ios::binary); f.write(reinterpret_cast<const char*>(id), sizeof id); }
        std::rename((id_path + ".tmp").c_str(), id_path.c_str());
    } else {
        for (int tries = 0;; ++tries) {
            std::ifstream f(id_path, std::ios::binary);
            if (f && f.read(reinterpret_cast<char*>(id), sizeof id)) break;
            if (tries > 600) { std::fprintf(stderr, "rank %d: no id file\n", rank); return 2; }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    const int device = std::getenv("TRGL_DEVICE") ? std::atoi(std::getenv("TRGL_DEVICE")) : rank;     // one GPU per rank
    void* comm = nullptr;
    CHK(trgl_rccl_comm_create(id, rank, world, device, &comm));

    // ---- the frame
    CHK(trgl_create(device, W, H, bpp, &ctx));
    if (bands) CHK(trgl_set_interleave(ctx, 32, rank, world));
    else CHK(trgl_set_strip(ctx, rank * (H / world), (rank + 1) * (H / world)));
    CHK(trgl_clear(ctx, nullptr, std::numeric_limits<double>::infinity()));
    CHK(trgl_draw(ctx, TRGL_SHADER_FLAT, nullptr, clip.data(), nullptr, colors.data(), n, TRGL_MEM_HOST));
    CHK(trgl_gather(ctx, comm, rank, world, /* with_z */ 1));            // flush + in-place all-gathers on the context's stream
    std::vector<uint8_t> fb(size_t(W) * H * bpp); std::vector<double> z(size_t(W) * H);
    CHK(trgl_read_framebuffer(ctx, fb.data()));
    CHK(trgl_read_zbuffer(ctx, z.data()));
    trgl_stats st{};
    CHK(trgl_get_stats(ctx, &st));
    char line[512];
    trgl_format_stats(&st, line, sizeof line);
    std::ofstream out(std::string(argv[2]) + "." + std::to_string(rank), std::ios::binary);
    out.write(reinterpret_cast<const char*>(fb.data()), std::streamsize(fb.size()));
    out.write(reinterpret_cast<const char*>(z.data()), std::streamsize(z.size() * 8));
    out.write(line, std::streamsize(std::strlen(line)));
    trgl_destroy(ctx);
    trgl_rccl_comm_destroy(comm);
    return out ? 0 : 3;
}
