// launch.h — host-callable launchers of the gfx950 kernels (kernels_bin.hip, kernels_raster.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "trgl_device.h"

namespace trgl {

uint32_t setup_num_blocks(uint32_t n);      // blocks of 256 triangles of one draw (k_setup and k_expand use the same)
// (`draw` travels as a kernel argument; the kernel leaves it at draws_dev[draw_idx] for the kernels behind it)
void launch_setup(hipStream_t s, const FrameParams& fp, const DrawDesc& draw, DrawDesc* draws_dev, int draw_idx, uint32_t n,
                  TriRec* recs, TriW* recs_w, uint32_t* cnt, uint2* tilebox, DevStats* stats, uint32_t* blk_sums, uint32_t blk_base);
// chunk_off[c] = pairs before setup block 16c; *total64 = all pairs of the flush
// (host_copy: pinned host memory that receives the pair count and the two counts behind it in DevStats)
// zero, zero_bytes (a multiple of 16, 16-byte aligned): also cleared by the kernel (the tile bounds of the flush)
void launch_chunk_spine(hipStream_t s, const uint32_t* blk_sums, uint32_t nblk, uint32_t* chunk_off, unsigned long long* total64, unsigned long long* host_copy,
                        void* zero, size_t zero_bytes);


// expand / radix / bounds read the flush's pair count from device memory and cover `cap` (the capacity of the pair buffers)
// with their grids; they do nothing when the count exceeds it (the host then grows the buffers and queues them again)
void launch_expand(hipStream_t s, const FrameParams& fp, uint32_t first, uint32_t n, int tiles_x, const uint32_t* cnt, const uint32_t* blk_sums,
                   const uint32_t* chunk_off, uint32_t blk_base, const uint2* tilebox, void* keys, bool key16, uint32_t* vals, uint16_t* bmask,
                   const unsigned long long* pairs_total, uint32_t cap);

uint32_t radix_num_workers(uint32_t P);
// key16: the keys (tile indices) are 16-bit words (frames of at most 65536 tiles), else 32-bit
// (the 4x4 block mask of every pair travels with it)
void launch_radix_pass(hipStream_t s, const void* keys_in, const uint32_t* vals_in, const uint16_t* msk_in, void* keys_out,
                       uint32_t* vals_out, uint16_t* msk_out, bool key16, const unsigned long long* pairs_total, uint32_t cap, int shift, int bits,
                       uint32_t* hist, uint32_t* scan_tmp);

void launch_bounds(hipStream_t s, const void* keys, bool key16, const unsigned long long* pairs_total, uint32_t cap,
                   uint32_t* tile_start, uint32_t* tile_end);

uint32_t owned_tiles(const FrameParams& fp);       // tiles of the rows this context owns (strip or interleaved bands)
uint32_t raster_max_items(const FrameParams& fp);   // work items (workgroups of k_raster) of a flush, at most
// item_stats: 4 x uint64 per work item (one partial of the counters per workgroup)
void launch_raster(hipStream_t s, const FrameParams& fp, int kind, bool all_well_scaled, const TriRec* recs, const TriW* recs_w,
                   const uint32_t* vals, const uint16_t* bmask,
                   const uint32_t* tile_start, const uint32_t* tile_end, const DrawDesc* draws,
                   const DevTexture* tex, DevStats* stats, uint32_t max_items, uint4* items,
                   uint32_t* n_items, unsigned long long* item_stats, hipEvent_t ev_before = nullptr,
                   hipEvent_t ev_after = nullptr);     // optional events recorded right around the k_raster launch

void launch_vertex_stage(hipStream_t s, const double mv[16], const double proj[16], const double* vertices, int stride,
                         const uint32_t* indices, uint32_t nfaces, double* clip, double* vary);
void launch_zimage(hipStream_t s, const double* zb, int W, int H, unsigned long long* keys2, uint8_t* out);
void launch_ssao(hipStream_t s, const double* zb, int W, int H, const double* dir_x, const double* dir_y, int ndir, int steps,
                 double radius, double threshold, double intensity, uint8_t* out);
void launch_composite(hipStream_t s, const uint8_t* fb, int bpp, const uint8_t* ao, int W, int H, uint8_t* out);

void launch_selftest_sampler(hipStream_t s, const DevTexture* tex, int slot, const double* uv, unsigned long long n, uint8_t* out);
void launch_selftest_division(hipStream_t s, unsigned long long n_per_thread, unsigned long long seed,
                              unsigned long long* mismatches);

}  // namespace trgl
