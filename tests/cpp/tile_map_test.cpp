// CPU-only unit test of the shard slot maps (jtx_tiles.hpp) that the multi-device exchange relies on:
// for ragged frame sizes and several shard counts, slot -> pixel and pixel -> slot are inverse bijections between
// "owned pixels of rank r" and "valid slots of rank r", every pixel has exactly one owner, and the counts match.
#include "../../jtx-pathtracer_amd/csrc/jtx_tiles.hpp"
#include <cstdio>
#include <vector>

int main() {
    const int sizes[][2] = {{1920, 1080}, {3840, 2160}, {512, 512}, {33, 65}, {1, 1}, {96, 64}, {31, 200}};
    const int worlds[] = {1, 2, 3, 4, 7, 8};
    long checked = 0;
    for (auto &sz : sizes) for (int world : worlds) {
        const int W = sz[0], H = sz[1];
        std::vector<int> owner((size_t) W * H, -1);
        int tilesSum = 0;
        for (int rank = 0; rank < world; ++rank) {
            const int owned = jtx::ownedTiles(W, H, rank, world);
            tilesSum += owned;
            long valid = 0;
            for (int slot = 0; slot < owned * 1024; ++slot) {
                int row, col;
                if (!jtx::slotToPixel(slot, rank, world, W, H, row, col)) continue;
                ++valid;
                if (owner[(size_t) row * W + col] != -1) { std::printf("pixel (%d,%d) owned twice (%dx%d world %d)\n", row, col, W, H, world); return 1; }
                owner[(size_t) row * W + col] = rank;
                int r2, s2; jtx::pixelToSlot(row, col, world, W, r2, s2);
                if (r2 != rank || s2 != slot) { std::printf("pixelToSlot != inverse at (%d,%d): rank %d/%d slot %d/%d\n", row, col, r2, rank, s2, slot); return 1; }
            }
            checked += valid;
        }
        if (tilesSum != jtx::tilesTotal(W, H)) { std::printf("tile counts do not add up\n"); return 1; }
        for (int v : owner) if (v < 0) { std::printf("unowned pixel (%dx%d world %d)\n", W, H, world); return 1; }
    }
    std::printf("tile maps ok (%ld pixel-slot pairs)\n", checked);
    return 0;
}
