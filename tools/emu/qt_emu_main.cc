// Runs the quadtree kernel of ovs_orb.hip (cut out by gen_quadtree_emu.py into qt_kernel.inc) for one
// frame on the host-side SIMT emulator, from per-level candidate lists, with every buffer the kernel
// indexes in guarded memory filled with a chosen byte.
//
//   qt_emu <in.bin> <out.bin> [fill byte, default 0xFF] [strict]
//   in.bin  (int32): L, then per level: w, h, quota, n, n x (x, y, response)   (x, y relative to the 19-px border)
//   out.bin (int32): per level: count, count x (x | y << 16, response)
//
// Built with -DQT_PATCHED against a kernel file that has the NODES_LDS template parameter and the
// (level0, ncap, node_ws) arguments, the launch is split as the experiment's host code split it:
// levels whose node arrays exceed 160 KB of LDS go to the instance that keeps them in a global workspace.
#include "simt_emu.h"

static u32* g_smem = nullptr;

#include "qt_kernel.inc"

#include <cmath>

static size_t lds_bytes(const plan_dev& p, int ncap) {
    return static_cast<size_t>(ncap) * 72 + static_cast<size_t>(p.max_cells + 1) * 4 + (16 + 8) * 4;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: qt_emu in.bin out.bin [fill]\n");
        return 1;
    }
    const int fill = argc > 3 ? (int)strtol(argv[3], nullptr, 0) : 0xFF;
    // reads past the LDS allocation: 0 as on the hardware (default), or a fault with "strict" as the fourth argument
    const bool lds_hw = !(argc > 4 && !strcmp(argv[4], "strict"));
    emu::install_segv_handler();
    FILE* f = fopen(argv[1], "rb");
    if (!f) {
        perror(argv[1]);
        return 1;
    }
    std::vector<int> in;
    {
        int v;
        while (fread(&v, 4, 1, f) == 1) {
            in.push_back(v);
        }
        fclose(f);
    }
    size_t pos = 0;
    const int L = in[pos++];
    plan_dev p;
    memset(&p, 0, sizeof(p));
    p.num_levels = L;
    std::vector<std::vector<int>> cand(L);
    int cell_base = 0, kp_base = 0;
    long long cand_base = 0;
    for (int l = 0; l < L; ++l) {
        level_geom& g = p.lv[l];
        g.w = in[pos++];
        g.h = in[pos++];
        g.quota = in[pos++];
        const int n = in[pos++];
        cand[l].assign(in.begin() + pos, in.begin() + pos + 3 * n);
        pos += 3 * n;
        const unsigned bw = g.w - 2 * kBorder, bh = g.h - 2 * kBorder;
        g.cells_x = (int)(bw / kCell) + 1;
        g.cells_y = (int)(bh / kCell) + 1;
        g.cell_base = cell_base;
        cell_base += g.cells_x * g.cells_y;
        const double ratio = (double)bw / bh;
        if (ratio > 1) {
            g.nx = (int)(unsigned)std::round(ratio);
            g.ny = 1;
            g.delta_x = (double)bw / g.nx;
            g.delta_y = bh;
        }
        else {
            g.nx = 1;
            g.ny = (int)(unsigned)std::round(1 / ratio);
            g.delta_x = bw;
            g.delta_y = (double)bh / g.ny;
        }
        g.ncap = std::max(2 * g.quota, 4 * g.nx * g.ny) + 8;
        g.kp_base = kp_base;
        kp_base += g.ncap;
        g.cand_base = cand_base;
        cand_base += (long long)g.cells_x * g.cells_y * kSlotsPerCell;
        p.max_cells = std::max(p.max_cells, g.cells_x * g.cells_y);
        p.max_ncap = std::max(p.max_ncap, g.ncap);
    }
    p.total_cells = cell_base;
    p.kp_total_cap = kp_base;
    p.frame_cand_cap = cand_base;

    u32* slots = static_cast<u32*>(emu::guarded_alloc((size_t)p.total_cells * kSlotsPerCell * 4, "slots", fill));
    u32* cell_counts = static_cast<u32*>(emu::guarded_alloc((size_t)p.total_cells * 4, "cell_counts", 0));
    for (int l = 0; l < L; ++l) {
        const level_geom& g = p.lv[l];
        int last_cell = -1;
        for (size_t i = 0; i + 2 < cand[l].size(); i += 3) {
            const int x = cand[l][i], y = cand[l][i + 1], r = cand[l][i + 2];
            const int cx = (x - 3) / kCell, cy = (y - 3) / kCell;
            if (cx >= g.cells_x || cy >= g.cells_y || x < 3 || y < 3) {
                fprintf(stderr, "level %d: candidate (%d, %d) outside the cell grid\n", l, x, y);
                return 1;
            }
            const int c = cy * g.cells_x + cx;
            if (c < last_cell) {
                fprintf(stderr, "level %d: candidates not in cell order\n", l);
                return 1;
            }
            last_cell = c;
            u32& n = cell_counts[g.cell_base + c];
            if (n >= (u32)kSlotsPerCell) {
                fprintf(stderr, "level %d: more than %d candidates in a cell\n", l, kSlotsPerCell);
                return 1;
            }
            slots[(size_t)(g.cell_base + c) * kSlotsPerCell + n] = (u32)x | ((u32)y << 12) | ((u32)r << 24);
            ++n;
        }
    }
    u32* cand_xy = static_cast<u32*>(emu::guarded_alloc((size_t)p.frame_cand_cap * 4, "cand_xy", fill));
    u32* cand_node = static_cast<u32*>(emu::guarded_alloc((size_t)p.frame_cand_cap * 4, "cand_node", fill));
    u32* lvl_xy = static_cast<u32*>(emu::guarded_alloc((size_t)p.kp_total_cap * 4, "lvl_xy", fill));
    uint8_t* lvl_resp = static_cast<uint8_t*>(emu::guarded_alloc((size_t)p.kp_total_cap, "lvl_resp", fill));
    int* lvl_count = static_cast<int*>(emu::guarded_alloc((size_t)(L + 16) * 4, "lvl_count", fill));

    emu_dim3 grid, blk;
#ifdef QT_PATCHED
    int first_lds = 0;
    for (int l = 0; l < L; ++l) {
        if (lds_bytes(p, p.lv[l].ncap) > 160 * 1024) {
            first_lds = l + 1;
        }
    }
    fprintf(stderr, "levels [0, %d) keep their nodes in the workspace, [%d, %d) in LDS\n", first_lds, first_lds, L);
    if (first_lds > 0) {
        int ncap = 0;
        for (int l = 0; l < first_lds; ++l) {
            ncap = std::max(ncap, p.lv[l].ncap);
        }
        u32* ws = static_cast<u32*>(emu::guarded_alloc((size_t)first_lds * ncap * kQtWordsPerNode * 4, "node workspace", fill));
        g_smem = static_cast<u32*>(emu::guarded_alloc((size_t)(p.max_cells + 1) * 4 + (16 + 8) * 4, "LDS of the workspace instance", fill, lds_hw));
        grid.x = 1;
        grid.y = first_lds;
        for (int l = 0; l < first_lds; ++l) {
            blk.x = 0;
            blk.y = l;
            emu::run_workgroup(kQtThreads, blk, grid, [&] {
                k_quadtree<false>(p, slots, cell_counts, cand_xy, cand_node, lvl_xy, lvl_resp, lvl_count, 0, ncap, ws);
            });
            fprintf(stderr, "level %d (workspace): count %d\n", l, lvl_count[l]);
        }
    }
    if (first_lds < L) {
        int ncap = 0;
        for (int l = first_lds; l < L; ++l) {
            ncap = std::max(ncap, p.lv[l].ncap);
        }
        g_smem = static_cast<u32*>(emu::guarded_alloc(lds_bytes(p, ncap), "LDS of the LDS instance", fill, lds_hw));
        grid.x = 1;
        grid.y = L - first_lds;
        for (int l = first_lds; l < L; ++l) {
            blk.x = 0;
            blk.y = l - first_lds;
            emu::run_workgroup(kQtThreads, blk, grid, [&] {
                k_quadtree<true>(p, slots, cell_counts, cand_xy, cand_node, lvl_xy, lvl_resp, lvl_count, first_lds, ncap, nullptr);
            });
            fprintf(stderr, "level %d (LDS): count %d\n", l, lvl_count[l]);
        }
    }
#else
    g_smem = static_cast<u32*>(emu::guarded_alloc(lds_bytes(p, p.max_ncap), "LDS", fill, lds_hw));
    grid.x = 1;
    grid.y = L;
    for (int l = 0; l < L; ++l) {
        blk.x = 0;
        blk.y = l;
        emu::run_workgroup(kQtThreads, blk, grid, [&] { k_quadtree(p, slots, cell_counts, cand_xy, cand_node, lvl_xy, lvl_resp, lvl_count); });
        fprintf(stderr, "level %d: count %d\n", l, lvl_count[l]);
    }
#endif
    fprintf(stderr, "%lld barriers, %lld wave operations\n", emu::g_barriers, emu::g_waveops);

    FILE* o = fopen(argv[2], "wb");
    for (int l = 0; l < L; ++l) {
        const int n = lvl_count[l];
        fwrite(&n, 4, 1, o);
        for (int q = 0; q < n; ++q) {
            const int xy = (int)lvl_xy[p.lv[l].kp_base + q], r = lvl_resp[p.lv[l].kp_base + q];
            fwrite(&xy, 4, 1, o);
            fwrite(&r, 4, 1, o);
        }
    }
    fclose(o);
    return 0;
}
