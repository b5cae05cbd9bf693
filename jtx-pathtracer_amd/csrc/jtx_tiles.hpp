// jtx_tiles.hpp -- pixel-tile ownership and the compact "slot" order of a shard (host + device, no HIP types).
//
// The reference cuts the frame into 32x32 tiles, row-major (camera.cpp:55-64).  With `world` shards, shard `rank`
// owns the tiles k with k % world == rank (interleaved: sky-heavy and geometry-heavy regions spread over all GPUs).
// Inside a shard the pixels are numbered by SLOT:  slot = owned * 1024 + sub * 64 + lane, where `owned` counts the
// shard's tiles, `sub` the 16 8x8 blocks of a tile (row-major) and `lane` the 64 pixels of a block (row-major): the
// order in which waves work on them, the layout of the per-path radiance records and of the exchange slabs.
#pragma once
#if defined(__HIPCC__)
#define JTX_HD __host__ __device__ inline
#else
#define JTX_HD inline
#endif

namespace jtx {

JTX_HD int tilesX(int width) { return (width + 31) / 32; }
JTX_HD int tilesTotal(int width, int height) { return tilesX(width) * ((height + 31) / 32); }
JTX_HD int ownedTiles(int width, int height, int rank, int world) {
    const int tiles = tilesTotal(width, height);
    return tiles > rank ? (tiles - rank + world - 1) / world : 0;
}
// slot of shard (rank, world) -> pixel; false for the padding slots of tiles that overhang the frame
JTX_HD bool slotToPixel(int slot, int rank, int world, int width, int height, int &row, int &col) {
    const int owned = slot >> 10, sub = (slot >> 6) & 15, lane = slot & 63;
    const int tile = rank + owned * world;
    const int tx = tilesX(width);
    const int trow = tile / tx, tcol = tile - trow * tx;
    row = trow * 32 + (sub >> 2) * 8 + (lane >> 3);
    col = tcol * 32 + (sub & 3) * 8 + (lane & 7);
    return row < height && col < width;
}
// pixel -> (owning rank, slot in that shard)
JTX_HD void pixelToSlot(int row, int col, int world, int width, int &rank, int &slot) {
    const int tile = (row >> 5) * tilesX(width) + (col >> 5);
    rank = tile % world;
    const int owned = tile / world;
    const int r = row & 31, c = col & 31;
    slot = owned * 1024 + ((r >> 3) * 4 + (c >> 3)) * 64 + (r & 7) * 8 + (c & 7);
}

} // namespace jtx
