/*
 * ky_shard.hpp -- how a frame shard becomes work items: the chunk schedule of a pixel's samples and the shard's tile / block geometry (ShardConst).
 * Host and device arithmetic only (KY_HD): the render kernel decodes work items with it (ky_render.hpp), the host sizes launches and validates
 * parameters with it (ky_pack.cpp, ky_launch.hip), and -- being plain C++ -- it is part of the sanitizer build (`make sanitize`).
 */
#pragma once
#include "ky_scene.hpp"

// Samples of a pixel are cut into chunks (= work items) by a schedule that depends on spp ONLY (chunk boundaries must not depend on
// the sharding, or images would differ between GPU counts: a chunk's samples are summed in float before they enter the fixed-point
// accumulator).  The bulk is KY_CHUNK-sample chunks; the END of the sample range tapers off -- KY_TAPER_16 samples in 16-sample
// chunks, then KY_TAPER_8 in 8s, KY_TAPER_4 in 4s, KY_TAPER_2 in 2s -- and items are queued chunk-major, so the small chunks of all
// blocks come last: the tail of a launch, where wavefronts run out of work one by one, is as long as ONE of the smallest items, while
// nearly all samples are rendered in chunks large enough to make the per-chunk bookkeeping (flush, refill) invisible.
// Sizing (tools/shard_scan.py, profiles/r03_taper_scan.txt): a wavefront needs about 1.1 ms for a 32-sample item, and wavefronts
// finish their last one up to that far apart; the stage that follows evens it out if it holds at least as much work per wavefront,
// which for a 1/8 shard of configs[1] (7.3 ms per launch) is 256 samples of 16, then 128 of 8, then 64 of 4.  Measured kernel-level
// efficiency at N = 8: no taper 0.88, 256 samples of 8 (round 2) 0.93, 128/64/32 0.92, 256/128/64 0.96; longer tapers (384/192/96/48)
// do not make the shard faster and cost the full frame 1-2.5 %.
#ifndef KY_CHUNK_BIG
#define KY_CHUNK_BIG 24   // round 4 (profiles/r04_b_chunk_scan.txt): 24-sample bulk chunks leave the full frame where 32 had it (50.5 against 50.6 ms) and make the
#endif                    // slowest 1/8 shard of configs[1] 2-3 % faster (6.65-6.68 against 6.79-6.89 ms: N = 8 kernel efficiency 0.947-0.950 against 0.918-0.931);
                          // 16 and 20 cost the full frame 1.3-1.9 %, and no other taper (192/96/48, a 2-sample stage, ...) beat 256/128/64 in shard time
#ifndef KY_TAPER_16
#define KY_TAPER_16 256
#endif
#ifndef KY_TAPER_8
#define KY_TAPER_8 128
#endif
#ifndef KY_TAPER_4
#define KY_TAPER_4 64
#endif
#ifndef KY_TAPER_2
#define KY_TAPER_2 0
#endif
constexpr int KY_CHUNK = KY_CHUNK_BIG;
static_assert(KY_CHUNK_BIG <= 127, "the lane's sample cursor keeps the chunk's remaining samples in 7 bits");
#ifndef KY_RING_SLOTS
#define KY_RING_SLOTS 3   // (a wave reads at most the two newest items; three slots keep the standard kernels' LDS block under 20 480 bytes: eight per CU)
#endif
constexpr int KY_RING = KY_RING_SLOTS;          // fetched-but-not-yet-started items a wave can hold
constexpr double KY_FIX_SCALE = 4294967296.0;   // 2^32: accumulator resolution 2.3e-10, range +-2.1e9

// The chunk schedule of `spp` samples (host and device; wave-uniform scalar arithmetic on the device, once per fetched item; written
// without arrays so that nothing of it lives in scratch memory).
struct ChunkPlan {
    int n_big, head;            // chunks of KY_CHUNK samples cover [0, head)
    int b1, b2, b3, b4;         // 16-sample chunks cover [head, b1), 8s [b1, b2), 4s [b2, b3), 2s [b3, b4 = spp)
    int n16, n8, n4, n2;
};
KY_HD inline ChunkPlan chunk_plan(int spp) {
    ChunkPlan p;
    p.b4 = spp;
    p.b3 = p.b4 > KY_TAPER_2 ? p.b4 - KY_TAPER_2 : 0;
    p.b2 = p.b3 > KY_TAPER_4 ? p.b3 - KY_TAPER_4 : 0;
    p.b1 = p.b2 > KY_TAPER_8 ? p.b2 - KY_TAPER_8 : 0;
    const int b0 = p.b1 > KY_TAPER_16 ? p.b1 - KY_TAPER_16 : 0;
    p.head = (b0 / KY_CHUNK) * KY_CHUNK;   // what is left of the bulk's last chunk goes to the 16-sample segment
    p.n_big = p.head / KY_CHUNK;
    p.n16 = (p.b1 - p.head + 15) / 16;
    p.n8 = (p.b2 - p.b1 + 7) / 8;
    p.n4 = (p.b3 - p.b2 + 3) / 4;
    p.n2 = (p.b4 - p.b3 + 1) / 2;
    return p;
}
KY_HD inline int chunk_count(const ChunkPlan& p) { return p.n_big + p.n16 + p.n8 + p.n4 + p.n2; }
KY_HD inline void chunk_range(const ChunkPlan& p, int c, int& s_begin, int& s_end) {
    int size = KY_CHUNK, first = 0, limit = p.head, n_seg = p.n_big;
    c -= p.n_big;
    if (c >= 0) { size = 16; first = p.head; limit = p.b1; n_seg = p.n16; c -= p.n16; }
    if (c >= 0) { size = 8; first = p.b1; limit = p.b2; n_seg = p.n8; c -= p.n8; }
    if (c >= 0) { size = 4; first = p.b2; limit = p.b3; n_seg = p.n4; c -= p.n4; }
    if (c >= 0) { size = 2; first = p.b3; limit = p.b4; n_seg = p.n2; c -= p.n2; }
    // c is now (index inside its segment) - (chunks of that segment): count back from the segment's chunk count
    s_begin = first + (c + n_seg) * size;
    s_end = s_begin + size < limit ? s_begin + size : limit;
}

struct ShardConst {
    int tile_w, tile_h, tile_first, tile_step;
    int tiles_x, tiles_y, n_tiles;     // tiles of the whole film / tiles owned by this shard
    int blocks_w, blocks_per_tile;     // 8x8 pixel blocks inside a tile
    int n_blocks;                      // n_tiles * blocks_per_tile
    int n_chunks;                      // chunk_count(chunk_plan(spp))
    unsigned n_items;                  // n_blocks * n_chunks
    int n_pix;                         // n_tiles * tile_w * tile_h
};
