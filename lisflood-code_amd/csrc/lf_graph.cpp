// lf_graph.cpp -- LDD raster -> adjacency -> routing orders -> engine layout (host side, O(N)).
//
// Replaces rebuildFlowMatrix / decodeFlowMatrix / streamLookups / upDownLookups / topoDistFromSea /
// _setRoutingOrders of the reference (kinematic_wave_parallel.py:59-106, 140-158;
// kinematic_wave_parallel_tools.py:111-130).  The reference builds the level sets with one np.unique
// per level (O(N*NL)); here a single breadth-first pass from the outlets yields the distances, the
// level sets and the engine's sweep layout at once.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>

#include "lf_blocks.h"
#include "lf_common.h"

namespace {

// keypad code -> direction index 0..7 (IX_ADDS row, kinematic_wave_parallel.py:49-51), 8 = no flow
const int kRowAdd[8] = {1, 1, 0, -1, -1, -1, 0, 1};
const int kColAdd[8] = {0, 1, 1, 1, 0, -1, -1, -1};

inline int decode(int code)
{
    switch (code) {
    case 2: return 0;
    case 3: return 1;
    case 6: return 2;
    case 9: return 3;
    case 8: return 4;
    case 7: return 5;
    case 4: return 6;
    case 1: return 7;
    default: return 8; // 5 = pit, 0 = sea; anything else is undefined in the reference -> no flow
    }
}

// codes_at(i): LDD code of raster cell i (row-major); is_land(i): land mask.
// virtual_down (optional, [N], -1 = none): for a pit u of this LDD, the pixel v it drains into in the UNCUT LDD
// (structures.py:44-61 turns the cells just upstream of a lake / reservoir into pits).  Such a link carries no
// router flow but the structure at v reads ChanQ(u) of the previous sub-step, so u is given the SAME level as v
// (a zero-length edge): the fused sub-step wavefront can then run the structure between two launches.
template <typename CodeAt, typename IsLand>
int build(int H, int W, CodeAt codes_at, IsLand is_land, bool all_land, lf_graph **out,
          const int64_t *virtual_down = nullptr)
{
    if (H <= 0 || W <= 0) return lf_set_error(LF_E_INVALID, "bad raster shape %d x %d", H, W);
    const int64_t HW = (int64_t)H * W;
    lf_graph *g = new (std::nothrow) lf_graph();
    static std::atomic<uint64_t> next_serial{1};
    if (g) g->serial = next_serial.fetch_add(1);
    if (!g) return lf_set_error(LF_E_INVALID, "out of memory");
    g->H = H;
    g->W = W;
    try {
        // 1. pixel ids (row-major over the land mask, add1.py:268-282)
        std::vector<int32_t> land_points;
        int64_t n = HW;
        if (!all_land) {
            land_points.assign(HW, -1);
            n = 0;
            for (int64_t i = 0; i < HW; ++i)
                if (is_land(i)) land_points[i] = (int32_t)n++;
        }
        if (n >= (int64_t)1 << 31) {
            delete g;
            return lf_set_error(LF_E_INVALID, "too many land pixels (%lld >= 2^31)", (long long)n);
        }
        g->N = n;
        // 2. downstream pixel of every pixel (kwpt.py:119-126): none if the target is off-grid or not land
        g->down.assign(n, -1);
        std::vector<int32_t> nups(n + 1, 0);
        {
            int64_t p = 0;
            for (int r = 0; r < H; ++r)
                for (int c = 0; c < W; ++c) {
                    const int64_t i = (int64_t)r * W + c;
                    if (!all_land && land_points[i] < 0) continue;
                    const int d = decode(codes_at(i, p));
                    if (d < 8) {
                        const int rr = r + kRowAdd[d], cc = c + kColAdd[d];
                        if (rr >= 0 && cc >= 0 && rr < H && cc < W) {
                            const int64_t j = (int64_t)rr * W + cc;
                            const int32_t dn = all_land ? (int32_t)j : land_points[j];
                            if (dn >= 0) {
                                g->down[p] = dn;
                                nups[dn]++;
                            }
                        }
                    }
                    ++p;
                }
        }
        std::vector<int32_t>().swap(land_points);
        int K = 0;
        for (int64_t p = 0; p < n; ++p) K = std::max(K, (int)nups[p]);
        g->K = std::max(1, K);
        // 3. upstream adjacency in pixel space (CSR, ascending source id as kwpt.py:127-128)
        std::vector<int32_t> uptr(n + 1);
        {
            int64_t acc = 0;
            for (int64_t p = 0; p < n; ++p) {
                uptr[p] = (int32_t)acc;
                acc += nups[p];
            }
            uptr[n] = (int32_t)acc;
        }
        std::vector<int32_t> uidx(uptr[n]);
        {
            std::vector<int32_t> fill(uptr.begin(), uptr.end() - 1);
            for (int64_t p = 0; p < n; ++p)
                if (g->down[p] >= 0) uidx[fill[g->down[p]]++] = (int32_t)p;
        }
        // 4. breadth-first search from the outlets (topoDistFromSea, kinematic_wave_parallel.py:92-106):
        //    generation k holds the pixels at distance k+1 from their outlet.
        std::vector<int32_t> queue(n);
        std::vector<int64_t> gen_start;
        int64_t tail = 0;
        // outlets WITH upstream cells first, isolated pixels (no upstream, no downstream: e.g. the non-channel land
        // pixels of the channel LDD) after them: they end up side by side at the end of the last level, where the
        // fused sub-step wavefront can skip them line by line while their state is zero
        if (!virtual_down) {
            for (int64_t p = 0; p < n; ++p)
                if (g->down[p] < 0 && nups[p] > 0) queue[tail++] = (int32_t)p;
            for (int64_t p = 0; p < n; ++p)
                if (g->down[p] < 0 && nups[p] == 0) queue[tail++] = (int32_t)p;
            int64_t head = 0;
            gen_start.push_back(0);
            while (head < tail) {
                const int64_t gen_end = tail;
                for (; head < gen_end; ++head) {
                    const int32_t p = queue[head];
                    for (int32_t e = uptr[p]; e < uptr[p + 1]; ++e) queue[tail++] = uidx[e];
                }
                gen_start.push_back(gen_end);
            }
        } else {
            // zero-length links: CSR of the pits hanging on every pixel, ascending id
            std::vector<int32_t> vptr(n + 1, 0);
            for (int64_t u = 0; u < n; ++u) {
                const int64_t v = virtual_down[u];
                if (v < 0) continue;
                if (v >= n || v == u || g->down[u] >= 0) {
                    delete g;
                    return lf_set_error(LF_E_INVALID, "virtual_down[%lld] = %lld: the source must be a pit of this LDD and "
                                        "the target another pixel", (long long)u, (long long)v);
                }
                vptr[v + 1]++;
            }
            for (int64_t p = 0; p < n; ++p) vptr[p + 1] += vptr[p];
            std::vector<int32_t> vidx(vptr[n] > 0 ? vptr[n] : 1), vfill(vptr.begin(), vptr.end() - 1);
            for (int64_t u = 0; u < n; ++u)
                if (virtual_down[u] >= 0) vidx[vfill[virtual_down[u]]++] = (int32_t)u;
            std::vector<int32_t> cur, next;
            for (int64_t p = 0; p < n; ++p)
                if (g->down[p] < 0 && virtual_down[p] < 0 && (nups[p] > 0 || vptr[p + 1] > vptr[p])) cur.push_back((int32_t)p);
            for (int64_t p = 0; p < n; ++p)
                if (g->down[p] < 0 && virtual_down[p] < 0 && nups[p] == 0 && vptr[p + 1] == vptr[p]) cur.push_back((int32_t)p);
            gen_start.push_back(0);
            while (!cur.empty() && tail < n) {
                for (size_t i = 0; i < cur.size() && (int64_t)(cur.size() + next.size()) <= n; ++i) { // cur grows
                    const int32_t p = cur[i];
                    for (int32_t e = uptr[p]; e < uptr[p + 1]; ++e) next.push_back(uidx[e]);
                    for (int32_t e = vptr[p]; e < vptr[p + 1]; ++e) cur.push_back(vidx[e]); // same generation
                }
                if (tail + (int64_t)cur.size() > n) break; // a cycle through a virtual link
                std::memcpy(queue.data() + tail, cur.data(), sizeof(int32_t) * cur.size());
                tail += (int64_t)cur.size();
                gen_start.push_back(tail);
                cur.swap(next);
                next.clear();
            }
        }
        if (tail != n) {
            delete g;
            return lf_set_error(LF_E_CYCLE, "LDD has a cycle: %lld of %lld pixels never reach an outlet",
                                (long long)(n - tail), (long long)n);
        }
        // gen_start = [0, |gen0|, |gen0|+|gen1|, ..., n]; drop a possible duplicated tail entry
        while (gen_start.size() >= 2 && gen_start[gen_start.size() - 1] == gen_start[gen_start.size() - 2])
            gen_start.pop_back();
        if (gen_start.back() != n) gen_start.push_back(n);
        const int64_t NL = (int64_t)gen_start.size() - 1;
        g->NL = NL;
        // 5. sweep layout: order k = generation NL-1-k (routing_order = max - distance, :149)
        g->level_start.assign(NL + 1, 0);
        for (int64_t k = 0; k < NL; ++k) {
            const int64_t gen = NL - 1 - k;
            g->level_start[k + 1] = g->level_start[k] + (gen_start[gen + 1] - gen_start[gen]);
        }
        g->perm.resize(n);
        for (int64_t k = 0; k < NL; ++k) {
            const int64_t gen = NL - 1 - k;
            std::memcpy(g->perm.data() + g->level_start[k], queue.data() + gen_start[gen],
                        sizeof(int32_t) * (size_t)(gen_start[gen + 1] - gen_start[gen]));
        }
        // 6. contiguous upstream ranges: the children of the i-th cell of generation g are consecutive
        //    in generation g+1, in ascending pixel id.
        g->ups_ptr.resize(n + 1);
        for (int64_t k = 0; k < NL; ++k) {
            int64_t acc = (k == 0) ? 0 : g->level_start[k - 1];
            for (int64_t p = g->level_start[k]; p < g->level_start[k + 1]; ++p) {
                g->ups_ptr[p] = (int32_t)acc;
                acc += nups[g->perm[p]];
            }
        }
        g->ups_ptr[n] = (int32_t)(NL >= 1 ? g->level_start[NL - 1] : 0);
        if (virtual_down) {
            g->linked.assign(n, 0);
            for (int64_t p = 0; p < n; ++p)
                if (virtual_down[g->perm[p]] >= 0) {
                    g->linked[p] = 1;
                    g->has_links = true;
                }
            if (!g->has_links) g->linked.clear();
        }
    } catch (const std::bad_alloc &) {
        delete g;
        return lf_set_error(LF_E_INVALID, "out of host memory while building the graph");
    }
    *out = g;
    return LF_OK;
}

} // namespace

extern "C" {

int lf_graph_create_ex(const double *ldd_codes, const uint8_t *land_mask, int H, int W, const int64_t *virtual_down,
                       lf_graph **out)
{
    if (!ldd_codes || !land_mask || !out) return lf_set_error(LF_E_INVALID, "null argument");
    auto code = [&](int64_t, int64_t p) {
        const double c = ldd_codes[p];
        return (c >= 0.0 && c <= 9.0 && c == std::floor(c)) ? (int)c : 0;
    };
    auto land = [&](int64_t i) { return land_mask[i] != 0; };
    return build(H, W, code, land, false, out, virtual_down);
}

int lf_graph_create(const double *ldd_codes, const uint8_t *land_mask, int H, int W, lf_graph **out)
{
    return lf_graph_create_ex(ldd_codes, land_mask, H, W, nullptr, out);
}

int lf_graph_create_raster(const uint8_t *ldd_raster, const uint8_t *land_mask, int H, int W, lf_graph **out)
{
    if (!ldd_raster || !out) return lf_set_error(LF_E_INVALID, "null argument");
    auto code = [&](int64_t i, int64_t) { return (int)ldd_raster[i]; };
    if (land_mask) {
        auto land = [&](int64_t i) { return land_mask[i] != 0; };
        return build(H, W, code, land, false, out);
    }
    auto land = [](int64_t) { return true; };
    return build(H, W, code, land, true, out);
}


void lf_graph_destroy(lf_graph *g) { delete g; }
int64_t lf_graph_num_pixels(const lf_graph *g) { return g ? g->N : -1; }
int64_t lf_graph_num_levels(const lf_graph *g) { return g ? g->NL : -1; }
int lf_graph_max_upstream(const lf_graph *g) { return g ? g->K : -1; }

// Shape of the block plan a router would sweep this graph with (lf_blocks.h; host only, no device needed): blocks of up
// to lmax consecutive levels of at most `wide` cells, cones of at most max_cone cells per level.
int lf_graph_block_plan_stats(const lf_graph *g, int lmax, int64_t wide, int max_cone, int64_t out[6])
{
    if (!g || !out || lmax < 1 || max_cone < 1) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int i = 0; i < 6; ++i) out[i] = 0;
    if (g->NL < 1) return LF_OK;
    lf_block_plan plan;
    try {
        lf_build_level_blocks(g->level_start, 0, g->NL, lmax, wide, max_cone,
                              [&](int64_t pos) { return (int64_t)g->ups_ptr[pos]; }, plan);
    } catch (const std::bad_alloc &) {
        return lf_set_error(LF_E_INVALID, "out of host memory while building the level blocks");
    }
    plan.level.push_back((int)g->NL);
    const int NB = (int)plan.level.size() - 1;
    out[0] = NB;
    for (int b = 0; b < NB; ++b) {
        const int k0 = plan.level[b], nl = plan.level[b + 1] - k0;
        if (nl < 2) continue;
        const int64_t cones = plan.row[b + 1] - plan.row[b] - 1;
        out[1] += 1;
        out[2] += cones;
        out[3] += cones * nl;
        out[4] += g->level_start[k0 + nl] - g->level_start[k0];
        out[5] = cones > out[5] ? cones : out[5];
    }
    return LF_OK;
}

// The order in which the n cones of one (block, sub-step) are handed to the workgroups of a launch whose first workgroup
// for them has linear id `first_linear_id` (lf_blocks.h: lf_xcd_contiguous, the function the kernel calls): out[i] = cone of
// launch position i.  Host only.
int lf_xcd_contiguous_order(int n, unsigned int first_linear_id, int32_t *out)
{
    if (n < 0 || (n > 0 && !out)) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int i = 0; i < n; ++i) out[i] = lf_xcd_contiguous(i, n, first_linear_id + (unsigned)i);
    return LF_OK;
}

// What the cone kernels take for granted about that plan, checked cell by cell on the host: the upstream range of every
// cell of a cone lies inside the cone's range of the level above (it is read from the LDS row of that level: slot =
// position - start of the range, at most max_cone slots), and holds at most 8 cells.
//   out[0] cells whose range leaves the cone's range above   out[1] largest slot index + 1 any cell reads
//   out[2] largest upstream count                             out[3] cells checked
int lf_graph_block_plan_check(const lf_graph *g, int lmax, int64_t wide, int max_cone, int64_t out[4])
{
    if (!g || !out || lmax < 1 || max_cone < 1) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int i = 0; i < 4; ++i) out[i] = 0;
    if (g->NL < 1) return LF_OK;
    lf_block_plan plan;
    try {
        lf_build_level_blocks(g->level_start, 0, g->NL, lmax, wide, max_cone,
                              [&](int64_t pos) { return (int64_t)g->ups_ptr[pos]; }, plan);
    } catch (const std::bad_alloc &) {
        return lf_set_error(LF_E_INVALID, "out of host memory while building the level blocks");
    }
    plan.level.push_back((int)g->NL);
    const int NB = (int)plan.level.size() - 1;
    for (int b = 0; b < NB; ++b) {
        const int k0 = plan.level[b], nl = plan.level[b + 1] - k0;
        if (nl < 2) continue;
        const int64_t cones = plan.row[b + 1] - plan.row[b] - 1;
        const int *rows = plan.cone.data() + plan.off[b];
        for (int64_t c = 0; c < cones; ++c) {
            const int *c0 = rows + c * nl, *c1 = c0 + nl;
            for (int j = 1; j < nl; ++j) {
                const int64_t above0 = c0[j - 1], above1 = c1[j - 1];
                if (above1 - above0 > max_cone) ++out[0];
                for (int64_t p = c0[j]; p < c1[j]; ++p) {
                    // (a level's last cell: its range runs on over the structure links parked at the end of the level above)
                    int64_t u0 = g->ups_ptr[p], u1 = g->ups_ptr[p + 1];
                    if (g->has_links) while (u1 > u0 && g->linked[u1 - 1]) --u1;
                    ++out[3];
                    if (u1 > u0 && (u0 < above0 || u1 > above1)) ++out[0];
                    if (u1 > u0) out[1] = std::max(out[1], u1 - above0);
                    out[2] = std::max(out[2], u1 - u0);
                }
            }
        }
    }
    return LF_OK;
}

int lf_graph_get_lookups(const lf_graph *g, double *downstream, int64_t *upstream, int64_t *num_upstream)
{
    if (!g || !downstream || !upstream || !num_upstream) return lf_set_error(LF_E_INVALID, "null argument");
    const int64_t n = g->N;
    const int K = g->K;
    for (int64_t p = 0; p < n; ++p) {
        downstream[p] = (double)g->down[p];
        num_upstream[p] = 0;
        for (int k = 0; k < K; ++k) upstream[p * K + k] = -1;
    }
    for (int64_t p = 0; p < n; ++p) { // ascending source id
        const int32_t d = g->down[p];
        if (d >= 0) upstream[(int64_t)d * K + num_upstream[d]++] = p;
    }
    return LF_OK;
}

int lf_graph_get_orders(const lf_graph *g, int64_t *pixels_ordered, int64_t *order_start_stop)
{
    if (!g || !pixels_ordered || !order_start_stop) return lf_set_error(LF_E_INVALID, "null argument");
    for (int64_t k = 0; k < g->NL; ++k) {
        const int64_t a = g->level_start[k], b = g->level_start[k + 1];
        order_start_stop[2 * k] = a;
        order_start_stop[2 * k + 1] = b;
        for (int64_t p = a; p < b; ++p) pixels_ordered[p] = g->perm[p];
        std::sort(pixels_ordered + a, pixels_ordered + b); // (order, pixel id), kinematic_wave_parallel.py:152
    }
    return LF_OK;
}

int lf_graph_get_layout(const lf_graph *g, int32_t *perm, int32_t *ups_ptr, int64_t *level_start)
{
    if (!g) return lf_set_error(LF_E_INVALID, "null argument");
    if (perm) std::memcpy(perm, g->perm.data(), sizeof(int32_t) * g->perm.size());
    if (ups_ptr) std::memcpy(ups_ptr, g->ups_ptr.data(), sizeof(int32_t) * g->ups_ptr.size());
    if (level_start) std::memcpy(level_start, g->level_start.data(), sizeof(int64_t) * g->level_start.size());
    return LF_OK;
}

/* linked[N] by POSITION: 1 = zero-length structure link (all 0 for a graph without links) */
int lf_graph_get_links(const lf_graph *g, uint8_t *linked)
{
    if (!g || !linked) return lf_set_error(LF_E_INVALID, "null argument");
    if (g->has_links)
        std::memcpy(linked, g->linked.data(), (size_t)g->N);
    else
        std::memset(linked, 0, (size_t)g->N);
    return LF_OK;
}

} // extern "C"
