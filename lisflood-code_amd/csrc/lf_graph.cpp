// lf_graph.cpp -- LDD raster -> adjacency -> routing orders -> engine layout (host side, O(N)).
//
// Replaces rebuildFlowMatrix / decodeFlowMatrix / streamLookups / upDownLookups / topoDistFromSea /
// _setRoutingOrders of the reference (kinematic_wave_parallel.py:59-106, 140-158;
// kinematic_wave_parallel_tools.py:111-130).  The reference builds the level sets with one np.unique
// per level (O(N*NL)); here a single breadth-first pass from the outlets yields the distances, the
// level sets and the engine's sweep layout at once.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

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

// ------------------------------------------------------------------------------------------------------------------
// Component layout: see lf_comp_plan in lf_common.h.  Host side, O(N * tiers) with the alive list compacted per tier.
// ------------------------------------------------------------------------------------------------------------------
int lf_graph_build_components(lf_graph *g, int64_t cap, int64_t bin_cells)
{
    if (!g) return lf_set_error(LF_E_INVALID, "null argument");
    if (g->has_links)
        return lf_set_error(LF_E_INVALID, "the component layout is not defined on a graph with structure links");
    // defaults from A/B runs on MI355X (deep 4000^2 / 10000^2, shallow 4000^2 / 10000^2): a tree bound that grows with
    // the domain (fewer tiers: the tiers run one after the other) and small bins (more wavefronts in flight)
    if (cap <= 0) {
        const int64_t want = std::min<int64_t>(std::max<int64_t>(g->N / 1024, 2048), 32768);
        cap = 2048;
        while (cap * 2 <= want) cap *= 2;
    }
    if (bin_cells <= 0) bin_cells = 2048;
    if (cap > (int64_t)1 << 30) cap = (int64_t)1 << 30;
    const int64_t n = g->N;
    lf_comp_plan *cp = new (std::nothrow) lf_comp_plan();
    if (!cp) return lf_set_error(LF_E_INVALID, "out of memory");
    try {
        cp->cap = cap;
        cp->bin_cells = bin_cells;
        cp->perm.resize(n);
        cp->ups_ptr.assign(n + 1, 0);
        cp->tier_bin_start.push_back(0);
        // upstream adjacency in pixel space, ascending source id
        std::vector<int32_t> uptr(n + 1, 0);
        for (int64_t p = 0; p < n; ++p)
            if (g->down[p] >= 0) uptr[g->down[p] + 1]++;
        for (int64_t p = 0; p < n; ++p) uptr[p + 1] += uptr[p];
        std::vector<int32_t> uidx(std::max<int64_t>(uptr[n], 1));
        {
            std::vector<int32_t> fill(uptr.begin(), uptr.end() - 1);
            for (int64_t p = 0; p < n; ++p)
                if (g->down[p] >= 0) uidx[fill[g->down[p]]++] = (int32_t)p;
        }
        std::vector<int16_t> tier(n, -1);
        std::vector<int32_t> size(n, 0), depth(n, 0), pos(n, -1);
        std::vector<int32_t> alive(g->perm.begin(), g->perm.end()); // topological order: upstream first
        std::vector<int32_t> roots, queue, next_alive;
        std::vector<int64_t> gen_start;
        int64_t cursor = 0; // next free position
        int t = 0;
        while (!alive.empty()) {
            if (t >= 32000) {
                delete cp;
                return lf_set_error(LF_E_INVALID, "component layout: more than 32000 tiers (cap too small for this LDD)");
            }
            // 1. sizes and depths of the trees of the cells still alive; tier t = the cells with <= cap alive cells upstream
            roots.clear();
            next_alive.clear();
            for (int32_t pix : alive) {
                int64_t s = 1;
                int32_t d = 1;
                for (int32_t e = uptr[pix]; e < uptr[pix + 1]; ++e) {
                    const int32_t c = uidx[e];
                    if (tier[c] >= 0 && tier[c] < t) continue; // an earlier tier: not part of this tree
                    s += size[c];
                    if (tier[c] == t && depth[c] + 1 > d) d = depth[c] + 1;
                }
                if (s > cap) s = cap + 1;
                size[pix] = (int32_t)s;
                depth[pix] = d;
                if (s <= cap)
                    tier[pix] = (int16_t)t;
                else
                    next_alive.push_back(pix);
            }
            for (int32_t pix : alive)
                if (tier[pix] == t && (g->down[pix] < 0 || tier[g->down[pix]] != t)) roots.push_back(pix);
            // 2. deepest trees first (counting sort by depth, ties in topological order)
            {
                int32_t dmax = 0;
                for (int32_t r : roots) dmax = std::max(dmax, depth[r]);
                std::vector<int64_t> cnt((size_t)dmax + 2, 0);
                for (int32_t r : roots) cnt[dmax - depth[r] + 1]++;
                for (int32_t d = 0; d <= dmax; ++d) cnt[d + 1] += cnt[d];
                std::vector<int32_t> sorted(roots.size());
                for (int32_t r : roots) sorted[cnt[dmax - depth[r]]++] = r;
                roots.swap(sorted);
            }
            if (t == 1) cp->trunk_first = cursor;
            // 3. bins: consecutive trees until bin_cells cells; layout by breadth-first generations from the roots
            size_t ri = 0;
            while (ri < roots.size()) {
                size_t rj = ri;
                int64_t cells = 0;
                while (rj < roots.size() && (cells == 0 || cells + size[roots[rj]] <= bin_cells)) cells += size[roots[rj++]];
                queue.clear();
                gen_start.clear();
                for (size_t i = ri; i < rj; ++i) queue.push_back(roots[i]);
                gen_start.push_back(0);
                size_t head = 0;
                while (head < queue.size()) {
                    const size_t gen_end = queue.size();
                    for (; head < gen_end; ++head) {
                        const int32_t pix = queue[head];
                        for (int32_t e = uptr[pix]; e < uptr[pix + 1]; ++e)
                            if (tier[uidx[e]] == t) queue.push_back(uidx[e]);
                    }
                    gen_start.push_back((int64_t)gen_end);
                }
                const int nl = (int)gen_start.size() - 1; // generations = local levels
                cp->bin_lvl_off.push_back((int32_t)cp->lvl.size());
                cp->bin_nl.push_back(nl);
                int64_t at = cursor;
                for (int k = 0; k < nl; ++k) { // level k = generation nl-1-k
                    const int gen = nl - 1 - k;
                    cp->lvl.push_back((int32_t)at);
                    for (int64_t i = gen_start[gen]; i < gen_start[gen + 1]; ++i) {
                        cp->perm[at] = queue[i];
                        pos[queue[i]] = (int32_t)at;
                        ++at;
                    }
                }
                cp->lvl.push_back((int32_t)at);
                // same-tier upstream ranges tile [bin start, start of the last level) in position order
                int64_t acc = cursor;
                for (int64_t p = cursor; p < at; ++p) {
                    cp->ups_ptr[p] = (int32_t)acc;
                    const int32_t pix = cp->perm[p];
                    for (int32_t e = uptr[pix]; e < uptr[pix + 1]; ++e) acc += tier[uidx[e]] == t;
                }
                cursor = at;
                ri = rj;
            }
            cp->tier_bin_start.push_back((int32_t)cp->bin_nl.size());
            alive.swap(next_alive);
            ++t;
        }
        cp->ups_ptr[n] = (int32_t)n;
        if (t <= 1) cp->trunk_first = n;
        // upstream index lists of the cells of tier >= 1 (all their upstream cells, ascending pixel id)
        const int64_t nt = n - cp->trunk_first;
        cp->t_ptr.assign(nt + 1, 0);
        for (int64_t q = 0; q < nt; ++q) {
            const int32_t pix = cp->perm[cp->trunk_first + q];
            cp->t_ptr[q + 1] = cp->t_ptr[q] + (uptr[pix + 1] - uptr[pix]);
        }
        cp->t_idx.resize(std::max<int64_t>(cp->t_ptr[nt], 1));
        for (int64_t q = 0; q < nt; ++q) {
            const int32_t pix = cp->perm[cp->trunk_first + q];
            int32_t o = cp->t_ptr[q];
            for (int32_t e = uptr[pix]; e < uptr[pix + 1]; ++e) cp->t_idx[o++] = pos[uidx[e]];
        }
    } catch (const std::bad_alloc &) {
        delete cp;
        return lf_set_error(LF_E_INVALID, "out of host memory while building the component layout");
    }
    delete g->comp;
    g->comp = cp;
    return LF_OK;
}

/* stats[0] tiers, [1] bins, [2] cells of tier >= 1, [3] largest number of local levels of a bin, [4] sum over tiers of
 * the deepest bin (the dependent chain of a sweep), [5] cap, [6] bin_cells */
int lf_graph_component_stats(const lf_graph *g, int64_t stats[7])
{
    if (!g || !stats) return lf_set_error(LF_E_INVALID, "null argument");
    if (!g->comp) return lf_set_error(LF_E_INVALID, "the graph has no component layout");
    const lf_comp_plan &c = *g->comp;
    const int T = (int)c.tier_bin_start.size() - 1;
    stats[0] = T;
    stats[1] = (int64_t)c.bin_nl.size();
    stats[2] = g->N - c.trunk_first;
    int64_t deepest = 0, chain = 0;
    for (int t = 0; t < T; ++t) {
        int64_t d = 0;
        for (int32_t b = c.tier_bin_start[t]; b < c.tier_bin_start[t + 1]; ++b) d = std::max<int64_t>(d, c.bin_nl[b]);
        deepest = std::max(deepest, d);
        chain += d;
    }
    stats[3] = deepest;
    stats[4] = chain;
    stats[5] = c.cap;
    stats[6] = c.bin_cells;
    return LF_OK;
}

/* sizes[0] = tiers + 1, [1] = bins, [2] = entries of lvl, [3] = entries of t_ptr, [4] = entries of t_idx: the lengths
 * of the arrays lf_graph_get_components fills (any pointer may be NULL) */
int lf_graph_component_sizes(const lf_graph *g, int64_t sizes[5])
{
    if (!g || !sizes) return lf_set_error(LF_E_INVALID, "null argument");
    if (!g->comp) return lf_set_error(LF_E_INVALID, "the graph has no component layout");
    const lf_comp_plan &c = *g->comp;
    sizes[0] = (int64_t)c.tier_bin_start.size();
    sizes[1] = (int64_t)c.bin_nl.size();
    sizes[2] = (int64_t)c.lvl.size();
    sizes[3] = (int64_t)c.t_ptr.size();
    sizes[4] = (int64_t)c.t_idx.size();
    return LF_OK;
}

int lf_graph_get_components(const lf_graph *g, int32_t *tier_bin_start, int32_t *bin_lvl_off, int32_t *bin_nl, int32_t *lvl,
                            int32_t *t_ptr, int32_t *t_idx, int64_t *trunk_first)
{
    if (!g) return lf_set_error(LF_E_INVALID, "null argument");
    if (!g->comp) return lf_set_error(LF_E_INVALID, "the graph has no component layout");
    const lf_comp_plan &c = *g->comp;
    auto put = [](int32_t *dst, const std::vector<int32_t> &v) {
        if (dst && !v.empty()) std::memcpy(dst, v.data(), sizeof(int32_t) * v.size());
    };
    put(tier_bin_start, c.tier_bin_start);
    put(bin_lvl_off, c.bin_lvl_off);
    put(bin_nl, c.bin_nl);
    put(lvl, c.lvl);
    put(t_ptr, c.t_ptr);
    put(t_idx, c.t_idx);
    if (trunk_first) *trunk_first = c.trunk_first;
    return LF_OK;
}

lf_graph::~lf_graph() { delete comp; }

void lf_graph_destroy(lf_graph *g) { delete g; }
int64_t lf_graph_num_pixels(const lf_graph *g) { return g ? g->N : -1; }
int64_t lf_graph_num_levels(const lf_graph *g) { return g ? g->NL : -1; }
int lf_graph_max_upstream(const lf_graph *g) { return g ? g->K : -1; }

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
    if (g->comp) { // component layout: the engine order routers on this graph use (level_start stays the reference's)
        if (perm) std::memcpy(perm, g->comp->perm.data(), sizeof(int32_t) * g->comp->perm.size());
        if (ups_ptr) std::memcpy(ups_ptr, g->comp->ups_ptr.data(), sizeof(int32_t) * g->comp->ups_ptr.size());
        if (level_start) std::memcpy(level_start, g->level_start.data(), sizeof(int64_t) * g->level_start.size());
        return LF_OK;
    }
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
