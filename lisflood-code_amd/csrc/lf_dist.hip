// lf_dist.hip -- row-block partitioned kinematic-wave routing across the GPUs of one node.
//
// The catchment raster is split into contiguous row blocks, one per rank (one process per GPU).  D8 edges
// span at most one row, so a rank only ever needs discharge values of the row just above and just below
// its block ("ghosts").  A cell's upstream neighbour may live on the other rank at ANY depth of the flow
// path, so one halo exchange per call would lag cross-boundary inflow by a call and change the numerical
// scheme.  Instead every local cell gets a PHASE = the largest number of rank-boundary crossings on any
// flow path into it:
//     phase(c) = max( phase(local upstream u), phase_on_its_rank(ghost upstream g) + 1 )
// and a call is   for j in 0..nphases-1:  sweep the local cells of phase j;  exchange the boundary cells of
// phase j (RCCL Send/Recv with the rank above / below, empty messages skipped).
// The result is bit-identical to the single-domain sweep: same per-cell arithmetic, same upstream
// summation order (ghosts of the row above first, then local cells in ascending id, then the row below =
// ascending global pixel id, kinematic_wave_parallel_tools.py:57-58,119-129).
//
// Local sweep order: cells sorted by (phase, level, breadth-first rank) with level = max local distance-to-outlet -
// distance as in the single-domain layout, so siblings stay adjacent and upstream gathers nearly contiguous;
// upstream positions come from an index list (lf_sweep.h INDEXED) because they may sit in earlier phases or in
// the ghost slots appended after the N local cells of the discharge vector.  Ghost slots are ordered
// (side, phase, column) so that the values received in round j land in one contiguous range.
#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "lf_blocks.h"
#include "lf_fused.h"

// ------------------------------------------------------------------------------------------------
// host: local graph with halos
// ------------------------------------------------------------------------------------------------
struct lf_dist_graph {
    int H = 0, W = 0;
    int64_t N = 0;        // local land cells
    int K = 1;            // max in-degree incl. ghosts
    bool finalized = false;
    std::vector<int32_t> down;     // [N] local downstream id, -1 none (true outlet or export)
    std::vector<int32_t> uptr, uidx; // local upstream CSR (pixel space, ascending id)
    std::vector<int32_t> topo;     // local cells, upstream before downstream
    // ghosts: cells of the halo rows that drain into a local cell; [0] = row above, [1] = row below
    std::vector<int32_t> ghost_target[2]; // local id each ghost drains into (ascending column)
    std::vector<int32_t> ghost_phase[2];  // phase of the ghost on its own rank
    // exports: local cells draining into a land cell of a halo row
    std::vector<int32_t> export_cell[2];  // local id (ascending column)
    std::vector<int32_t> phase, height;   // [N]; height = level (max local distance-to-outlet - distance)
    int nphases = 1;                      // global number of phases (set by the host after the fixpoint)
    // finalized layout
    std::vector<int32_t> perm, pos;       // position <-> local id
    std::vector<int32_t> ups_ptr, ups_idx;
    std::vector<int32_t> ups_base;        // [N] first upstream position when they are consecutive, else -1 (-> ups_idx)
    int64_t n_noncontiguous = 0;
    std::vector<int64_t> level_start;     // launch units: runs of equal (phase, height)
    std::vector<int32_t> phase_level;     // [nphases+1] first launch unit of each phase
    // Inside a phase the cells that some export of the phase depends on (the exports and their same-phase upstream
    // closure: "boundary-critical") come FIRST, the rest ("bulk") after them.  The two parts are independent of each
    // other -- a cell has one downstream cell, so nothing outside the closure drains into it and nothing inside it
    // drains out except through the export -- which lets the halo exchange of round j start after part 0 of phase j
    // and run beside part 1 (lf_dist_router_route, second stream).  stage = 2 * phase + part.
    std::vector<uint8_t> crit;            // [N] by local id: 1 = boundary-critical
    std::vector<int32_t> stage_level;     // [2 * nphases + 1] first launch unit of every stage
    std::vector<int32_t> export_pos[2];   // positions of exports sorted by (phase, column)
    std::vector<int64_t> export_off[2];   // [nphases+1] offsets into export_pos per phase
    std::vector<int64_t> ghost_off[2];    // [nphases+1] offsets (within the side) of ghosts per phase
    int64_t ghost_base[2] = {0, 0};       // slot of the first ghost of each side (added to N)
    // ---- fused sub-steps on the partition (lf_dist_routing_substeps_fused) ----
    // A rank sweeps ONE PHASE for ALL sub-steps of a model step as a skewed wavefront over (level, sub-step), like the
    // single-domain lf_routing_substeps_fused; what crosses a phase or a rank boundary -- the router outputs of cells
    // whose downstream cell sits in a later phase or on another rank -- is kept for every sub-step in SLABS indexed
    // [slot][sub-step], so a round's halo is one contiguous block per neighbour and section.  Slots:
    //   [exports top | exports bottom | ghosts top | ghosts bottom | local cells feeding a later phase],
    // exports and ghosts ordered (phase, column) as in the per-call exchange.
    std::vector<int32_t> out_slot;        // [N] by position: slot the cell's router outputs are stored in, -1 none
    std::vector<int32_t> ups_idx_f;       // as ups_idx: a same-phase local position, or -(slot) - 1
    int64_t n_slots = 0, slot_export[2] = {0, 0}, slot_ghost[2] = {0, 0}, slot_xphase = 0;
    // level blocks and cones of every phase (lf_blocks.h; k_fused_cones<DIST>): blocks never span two phases
    lf_block_plan fplan;
    std::vector<int32_t> fplan_phase_block; // [nphases + 1] first block of every phase
    // the same for single router calls (k_sweep_cones<DIST>): blocks never span two STAGES (a phase's boundary-critical
    // part and its bulk are swept separately), one wavefront per cone, blocks of up to LF_ROUTE_LEVELS (256) units
    lf_block_plan rplan;
    std::vector<int32_t> rplan_stage_block; // [2 * nphases + 1] first block of every stage
};

namespace {

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
    default: return 8;
    }
}

void local_phases(lf_dist_graph *g)
{
    const int64_t n = g->N;
    std::fill(g->phase.begin(), g->phase.end(), 0);
    for (int side = 0; side < 2; ++side)
        for (size_t i = 0; i < g->ghost_target[side].size(); ++i) {
            int32_t &ph = g->phase[g->ghost_target[side][i]];
            ph = std::max(ph, g->ghost_phase[side][i] + 1);
        }
    for (int64_t t = 0; t < n; ++t) { // upstream before downstream
        const int32_t c = g->topo[t];
        const int32_t d = g->down[c];
        if (d >= 0) g->phase[d] = std::max(g->phase[d], g->phase[c]);
    }
}

} // namespace

extern "C" {

int lf_dist_graph_create(const uint8_t *ldd_local, const uint8_t *mask_local, int H, int W, const uint8_t *ldd_top,
                         const uint8_t *mask_top, const uint8_t *ldd_bottom, const uint8_t *mask_bottom,
                         lf_dist_graph **out)
{
    if (!ldd_local || !out || H <= 0 || W <= 0) return lf_set_error(LF_E_INVALID, "bad argument");
    const int64_t HW = (int64_t)H * W;
    lf_dist_graph *g = new lf_dist_graph();
    g->H = H;
    g->W = W;
    std::vector<int32_t> lp(HW, -1);
    int64_t n = 0;
    for (int64_t i = 0; i < HW; ++i)
        if (!mask_local || mask_local[i]) lp[i] = (int32_t)n++;
    g->N = n;
    g->down.assign(n, -1);
    std::vector<int32_t> nups(n + 1, 0);
    const bool has_side[2] = {ldd_top != nullptr, ldd_bottom != nullptr};
    const uint8_t *halo_ldd[2] = {ldd_top, ldd_bottom};
    const uint8_t *halo_mask[2] = {mask_top, mask_bottom};
    // local edges and exports
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) {
            const int64_t i = (int64_t)r * W + c;
            const int32_t p = lp[i];
            if (p < 0) continue;
            const int d = decode(ldd_local[i]);
            if (d >= 8) continue;
            const int rr = r + kRowAdd[d], cc = c + kColAdd[d];
            if (cc < 0 || cc >= W) continue;
            if (rr >= 0 && rr < H) {
                const int32_t dn = lp[(int64_t)rr * W + cc];
                if (dn >= 0) {
                    g->down[p] = dn;
                    nups[dn]++;
                }
            } else {
                const int side = rr < 0 ? 0 : 1;
                if (has_side[side] && (!halo_mask[side] || halo_mask[side][cc])) g->export_cell[side].push_back(p);
            }
        }
    // ghosts: halo cells draining into a local land cell
    std::vector<int32_t> nghost(n, 0);
    for (int side = 0; side < 2; ++side) {
        if (!has_side[side]) continue;
        const int target_row = side == 0 ? 0 : H - 1;
        const int want_dr = side == 0 ? 1 : -1;
        for (int c = 0; c < W; ++c) {
            if (halo_mask[side] && !halo_mask[side][c]) continue;
            const int d = decode(halo_ldd[side][c]);
            if (d >= 8 || kRowAdd[d] != want_dr) continue;
            const int cc = c + kColAdd[d];
            if (cc < 0 || cc >= W) continue;
            const int32_t t = lp[(int64_t)target_row * W + cc];
            if (t < 0) continue;
            g->ghost_target[side].push_back(t);
            g->ghost_phase[side].push_back(0);
            nghost[t]++;
        }
    }
    int K = 0;
    for (int64_t p = 0; p < n; ++p) K = std::max(K, nups[p] + nghost[p]);
    g->K = std::max(1, K);
    // local upstream CSR
    g->uptr.resize(n + 1);
    int64_t acc = 0;
    for (int64_t p = 0; p < n; ++p) {
        g->uptr[p] = (int32_t)acc;
        acc += nups[p];
    }
    g->uptr[n] = (int32_t)acc;
    g->uidx.resize(acc);
    {
        std::vector<int32_t> fill(g->uptr.begin(), g->uptr.end() - 1);
        for (int64_t p = 0; p < n; ++p)
            if (g->down[p] >= 0) g->uidx[fill[g->down[p]]++] = (int32_t)p;
    }
    // breadth-first search from the local outlets (true outlets + exports): generation = distance to the local
    // outlet.  Reversed, the queue is a topological order (upstream first) that is sorted by level
    // (= max distance - distance, as in the single-domain layout) and keeps the children of a cell adjacent.
    std::vector<int32_t> queue(n);
    g->height.assign(n, 0);
    int64_t head = 0, tail = 0;
    for (int64_t p = 0; p < n; ++p)
        if (g->down[p] < 0) queue[tail++] = (int32_t)p;
    int32_t gen = 0;
    std::vector<int64_t> gen_start(1, 0);
    while (head < tail) {
        const int64_t gen_end = tail;
        for (; head < gen_end; ++head) {
            const int32_t p = queue[head];
            g->height[p] = gen;
            for (int32_t e = g->uptr[p]; e < g->uptr[p + 1]; ++e) queue[tail++] = g->uidx[e];
        }
        gen_start.push_back(gen_end);
        ++gen;
    }
    if (tail != n) {
        delete g;
        return lf_set_error(LF_E_CYCLE, "LDD has a cycle inside the row block");
    }
    for (int64_t p = 0; p < n; ++p) g->height[p] = (gen - 1) - g->height[p]; // level: every local upstream cell is one lower
    // generations from the farthest to the outlets, each in its breadth-first order: the children of consecutive cells
    // are consecutive AND ascending in pixel id, so a cell's upstream positions are one ascending run
    g->topo.clear();
    g->topo.reserve(n);
    for (int32_t k = gen - 1; k >= 0; --k) g->topo.insert(g->topo.end(), queue.begin() + gen_start[k], queue.begin() + gen_start[k + 1]);
    g->phase.assign(n, 0);
    local_phases(g);
    *out = g;
    return LF_OK;
}

void lf_dist_graph_destroy(lf_dist_graph *g) { delete g; }
int64_t lf_dist_graph_num_pixels(const lf_dist_graph *g) { return g ? g->N : -1; }

// out: n_export_top, n_export_bottom, n_ghost_top, n_ghost_bottom
int lf_dist_graph_counts(const lf_dist_graph *g, int64_t out[4])
{
    if (!g || !out) return lf_set_error(LF_E_INVALID, "null argument");
    out[0] = (int64_t)g->export_cell[0].size();
    out[1] = (int64_t)g->export_cell[1].size();
    out[2] = (int64_t)g->ghost_target[0].size();
    out[3] = (int64_t)g->ghost_target[1].size();
    return LF_OK;
}

int lf_dist_graph_get_export_phases(const lf_dist_graph *g, int32_t *top, int32_t *bottom)
{
    if (!g) return lf_set_error(LF_E_INVALID, "null argument");
    int32_t *dst[2] = {top, bottom};
    for (int side = 0; side < 2; ++side)
        for (size_t i = 0; i < g->export_cell[side].size(); ++i) dst[side][i] = g->phase[g->export_cell[side][i]];
    return LF_OK;
}

// ghost phases = the export phases of the neighbouring ranks (column order is identical on both sides).
// Recomputes the local phases; *changed = 1 if any export phase changed (the fixpoint iteration continues).
int lf_dist_graph_set_ghost_phases(lf_dist_graph *g, const int32_t *top, const int32_t *bottom, int *changed)
{
    if (!g || !changed) return lf_set_error(LF_E_INVALID, "null argument");
    if (g->finalized) return lf_set_error(LF_E_INVALID, "graph already finalized");
    const int32_t *src[2] = {top, bottom};
    std::vector<int32_t> before[2];
    for (int side = 0; side < 2; ++side) {
        before[side].resize(g->export_cell[side].size());
        for (size_t i = 0; i < before[side].size(); ++i) before[side][i] = g->phase[g->export_cell[side][i]];
        for (size_t i = 0; i < g->ghost_phase[side].size(); ++i) g->ghost_phase[side][i] = src[side][i];
    }
    local_phases(g);
    *changed = 0;
    for (int side = 0; side < 2; ++side)
        for (size_t i = 0; i < before[side].size(); ++i)
            if (before[side][i] != g->phase[g->export_cell[side][i]]) *changed = 1;
    return LF_OK;
}

int lf_dist_graph_local_num_phases(const lf_dist_graph *g)
{
    if (!g) return -1;
    int m = 0;
    for (int32_t p : g->phase) m = std::max(m, p);
    for (int side = 0; side < 2; ++side)
        for (int32_t p : g->ghost_phase[side]) m = std::max(m, p + 1);
    return m + 1;
}

// nphases: the global maximum over ranks of lf_dist_graph_local_num_phases (all ranks run the same rounds)
int lf_dist_graph_finalize(lf_dist_graph *g, int nphases)
{
    if (!g) return lf_set_error(LF_E_INVALID, "null argument");
    if (nphases < lf_dist_graph_local_num_phases(g)) return lf_set_error(LF_E_INVALID, "nphases too small");
    const int64_t n = g->N;
    g->nphases = nphases;
    // boundary-critical cells: the exports and everything of their own phase that drains into them
    g->crit.assign(n, 0);
    for (int side = 0; side < 2; ++side)
        for (int32_t c : g->export_cell[side]) g->crit[c] = 1;
    for (int64_t t = n - 1; t >= 0; --t) { // downstream before upstream
        const int32_t c = g->topo[t], d = g->down[c];
        if (d >= 0 && g->crit[d] && g->phase[d] == g->phase[c]) g->crit[c] = 1;
    }
    auto stage_of = [&](int32_t c) { return 2 * g->phase[c] + (g->crit[c] ? 0 : 1); };
    const int nstages = 2 * nphases;
    // order by (stage, level, breadth-first rank): `topo` is already sorted by level with siblings adjacent, a stable
    // counting sort by stage on top of it keeps that inside every stage (the same-stage children of a cell stay one
    // ascending run: a cell's same-phase children are in its own part)
    const std::vector<int32_t> &tmp = g->topo;
    g->perm.resize(n);
    {
        std::vector<int64_t> cnt(nstages + 1, 0);
        for (int64_t p = 0; p < n; ++p) cnt[stage_of((int32_t)p) + 1]++;
        for (int j = 0; j < nstages; ++j) cnt[j + 1] += cnt[j];
        for (int64_t i = 0; i < n; ++i) g->perm[cnt[stage_of(tmp[i])]++] = tmp[i];
    }
    g->pos.resize(n);
    for (int64_t p = 0; p < n; ++p) g->pos[g->perm[p]] = (int32_t)p;
    // launch units (runs of equal stage and level), stage and phase boundaries
    g->level_start.clear();
    g->stage_level.assign(nstages + 1, 0);
    {
        int cur_stage = -1, cur_h = -1;
        for (int64_t p = 0; p < n; ++p) {
            const int32_t c = g->perm[p];
            const int st = stage_of(c);
            if (st != cur_stage || g->height[c] != cur_h) {
                for (int j = cur_stage + 1; j <= st; ++j) g->stage_level[j] = (int32_t)g->level_start.size();
                g->level_start.push_back(p);
                cur_stage = st;
                cur_h = g->height[c];
            }
        }
        for (int j = cur_stage + 1; j <= nstages; ++j) g->stage_level[j] = (int32_t)g->level_start.size();
        g->level_start.push_back(n);
    }
    g->phase_level.assign(nphases + 1, 0);
    for (int j = 0; j <= nphases; ++j) g->phase_level[j] = g->stage_level[2 * j];
    // ghost slots ordered (side, phase, column); exports ordered (phase, column)
    std::vector<int32_t> ghost_slot[2];
    int64_t slot = 0;
    for (int side = 0; side < 2; ++side) {
        const size_t m = g->ghost_target[side].size();
        g->ghost_base[side] = slot;
        g->ghost_off[side].assign(nphases + 1, 0);
        for (size_t i = 0; i < m; ++i) g->ghost_off[side][g->ghost_phase[side][i] + 1]++;
        for (int j = 0; j < nphases; ++j) g->ghost_off[side][j + 1] += g->ghost_off[side][j];
        ghost_slot[side].resize(m);
        std::vector<int64_t> fill(g->ghost_off[side].begin(), g->ghost_off[side].end() - 1);
        for (size_t i = 0; i < m; ++i) ghost_slot[side][i] = (int32_t)(slot + fill[g->ghost_phase[side][i]]++);
        slot += (int64_t)m;
        const size_t me = g->export_cell[side].size();
        g->export_off[side].assign(nphases + 1, 0);
        for (size_t i = 0; i < me; ++i) g->export_off[side][g->phase[g->export_cell[side][i]] + 1]++;
        for (int j = 0; j < nphases; ++j) g->export_off[side][j + 1] += g->export_off[side][j];
        g->export_pos[side].resize(me);
        std::vector<int64_t> efill(g->export_off[side].begin(), g->export_off[side].end() - 1);
        for (size_t i = 0; i < me; ++i) {
            const int32_t c = g->export_cell[side][i];
            g->export_pos[side][efill[g->phase[c]]++] = g->pos[c];
        }
    }
    // upstream index lists in ascending global pixel id: row above, local (ascending id), row below
    std::vector<std::vector<int32_t>> gl[2];
    std::vector<int32_t> gcount[2];
    for (int side = 0; side < 2; ++side) gcount[side].assign(n, 0);
    for (int side = 0; side < 2; ++side)
        for (int32_t t : g->ghost_target[side]) gcount[side][t]++;
    g->ups_ptr.resize(n + 1);
    int64_t e = 0;
    for (int64_t p = 0; p < n; ++p) {
        const int32_t c = g->perm[p];
        g->ups_ptr[p] = (int32_t)e;
        e += gcount[0][c] + (g->uptr[c + 1] - g->uptr[c]) + gcount[1][c];
    }
    g->ups_ptr[n] = (int32_t)e;
    g->ups_idx.assign(e, -1);
    {
        std::vector<int32_t> cursor(n);
        for (int64_t p = 0; p < n; ++p) cursor[g->perm[p]] = g->ups_ptr[p];
        for (size_t i = 0; i < g->ghost_target[0].size(); ++i) // row above, ascending column
            g->ups_idx[cursor[g->ghost_target[0][i]]++] = (int32_t)(n + ghost_slot[0][i]);
        for (int64_t c = 0; c < n; ++c)
            for (int32_t k = g->uptr[c]; k < g->uptr[c + 1]; ++k) g->ups_idx[cursor[c]++] = g->pos[g->uidx[k]];
        for (size_t i = 0; i < g->ghost_target[1].size(); ++i) // row below, ascending column
            g->ups_idx[cursor[g->ghost_target[1][i]]++] = (int32_t)(n + ghost_slot[1][i]);
    }
    // the (phase, level, breadth-first rank) order leaves the upstream positions of nearly every cell consecutive:
    // those cells read their inflow as a coalesced run, only the others (ghost or cross-phase inflow) go through the list
    g->ups_base.assign(n, 0);
    g->n_noncontiguous = 0;
    for (int64_t p = 0; p < n; ++p) {
        const int32_t a = g->ups_ptr[p], b = g->ups_ptr[p + 1];
        int32_t base = (b > a) ? g->ups_idx[a] : 0;
        for (int32_t k = a; k < b; ++k)
            if (g->ups_idx[k] != base + (k - a)) {
                base = -1;
                break;
            }
        g->ups_base[p] = base;
        g->n_noncontiguous += base < 0;
    }
    // ---- slab slots and upstream lists of the fused sub-step wavefront ----
    {
        const int64_t ne0 = (int64_t)g->export_pos[0].size(), ne1 = (int64_t)g->export_pos[1].size();
        const int64_t ng0 = (int64_t)g->ghost_target[0].size(), ng1 = (int64_t)g->ghost_target[1].size();
        g->slot_export[0] = 0;
        g->slot_export[1] = ne0;
        g->slot_ghost[0] = ne0 + ne1;
        g->slot_ghost[1] = ne0 + ne1 + ng0;
        g->slot_xphase = ne0 + ne1 + ng0 + ng1;
        g->out_slot.assign(n, -1);
        for (int side = 0; side < 2; ++side)
            for (size_t i = 0; i < g->export_pos[side].size(); ++i)
                g->out_slot[g->export_pos[side][i]] = (int32_t)(g->slot_export[side] + (int64_t)i);
        int64_t x = g->slot_xphase;
        for (int64_t p = 0; p < n; ++p) {
            const int32_t c = g->perm[p], d = g->down[c];
            if (d >= 0 && g->phase[d] > g->phase[c]) g->out_slot[p] = (int32_t)x++;
        }
        g->n_slots = x;
        if (x >= ((int64_t)1 << 31)) return lf_set_error(LF_E_INVALID, "too many slab slots");
        g->ups_idx_f.resize(g->ups_idx.size());
        for (int64_t p = 0; p < n; ++p) {
            const int32_t ph = g->phase[g->perm[p]];
            for (int32_t k = g->ups_ptr[p]; k < g->ups_ptr[p + 1]; ++k) {
                const int32_t e2 = g->ups_idx[k];
                if (e2 >= n) // ghost: ups_idx holds n + ghost slot (side, phase, column)
                    g->ups_idx_f[k] = -(int32_t)(g->slot_ghost[0] + (e2 - n)) - 1;
                else if (g->phase[g->perm[e2]] == ph)
                    g->ups_idx_f[k] = e2;
                else
                    g->ups_idx_f[k] = -g->out_slot[e2] - 1;
            }
        }
    }
    // ---- level blocks and cones of every phase ----
    {
        int lmax = 16;
        if (const char *e = std::getenv("LF_FUSED_LEVELS")) lmax = std::atoi(e);
        lmax = lmax < 1 ? 1 : (lmax > 64 ? 64 : lmax);
        if ((int64_t)n >= ((int64_t)1 << 29)) lmax = 1; // k_fused_cones: 32-bit byte offsets (see build_level_blocks)
        int64_t wide = 262144;
        if (const char *e = std::getenv("LF_FUSED_WIDE")) wide = std::atoll(e);
        g->fplan = lf_block_plan();
        g->fplan_phase_block.assign(nphases + 1, 0);
        try {
            // child[a] = first position of the unit before whose SAME-PHASE downstream cell is at or behind a; cells that
            // drain into a later phase or another rank sit between the runs and go with the run behind them
            std::vector<int32_t> child((size_t)n, 0);
            std::vector<int64_t> dfill;
            for (int j = 0; j < nphases; ++j)
                for (int64_t k = g->phase_level[j] + 1; k < g->phase_level[j + 1]; ++k) {
                    const int64_t pb = g->level_start[k - 1], pe = g->level_start[k], e = g->level_start[k + 1];
                    dfill.assign((size_t)(pe - pb), 0);
                    int64_t nxt = e;
                    for (int64_t u = pe - 1; u >= pb; --u) {
                        const int32_t c = g->perm[u], d = g->down[c];
                        if (d >= 0 && g->phase[d] == g->phase[c] && g->pos[d] >= pe && g->pos[d] < e) nxt = g->pos[d];
                        dfill[(size_t)(u - pb)] = nxt;
                    }
                    int64_t u = pb;
                    for (int64_t a = pe; a < e; ++a) {
                        while (u < pe && dfill[(size_t)(u - pb)] < a) ++u;
                        child[(size_t)a] = (int32_t)u;
                    }
                }
            if (lmax > 1 && n < ((int64_t)1 << 31))
                for (int j = 0; j < nphases; ++j) {
                    g->fplan_phase_block[j] = (int32_t)g->fplan.level.size();
                    lf_build_level_blocks(g->level_start, g->phase_level[j], g->phase_level[j + 1], lmax, wide, kBlock,
                                          [&](int64_t pos) { return (int64_t)child[(size_t)pos]; }, g->fplan);
                }
            int rmax = 256;
            if (const char *e = std::getenv("LF_ROUTE_LEVELS")) rmax = std::atoi(e);
            rmax = rmax < 1 ? 1 : (rmax > 512 ? 512 : rmax);
            g->rplan = lf_block_plan();
            g->rplan_stage_block.assign(nstages + 1, 0);
            if (rmax > 1 && n < ((int64_t)1 << 31))
                for (int st = 0; st < nstages; ++st) {
                    g->rplan_stage_block[st] = (int32_t)g->rplan.level.size();
                    lf_build_level_blocks(g->level_start, g->stage_level[st], g->stage_level[st + 1], rmax, wide, 64,
                                          [&](int64_t pos) { return (int64_t)child[(size_t)pos]; }, g->rplan);
                }
            g->rplan_stage_block[nstages] = (int32_t)g->rplan.level.size();
            g->rplan.level.push_back((int)(g->level_start.size() - 1));
            if (!g->rplan.any_multi || g->rplan.cone.size() >= ((size_t)1 << 31)) { // one launch per unit it is
                g->rplan = lf_block_plan();
                g->rplan_stage_block.clear();
            }
        } catch (const std::bad_alloc &) {
            return lf_set_error(LF_E_INVALID, "out of host memory while building the level blocks");
        }
        g->fplan_phase_block[nphases] = (int32_t)g->fplan.level.size();
        g->fplan.level.push_back((int)(g->level_start.size() - 1));
        if (!g->fplan.any_multi || g->fplan.cone.size() >= ((size_t)1 << 31)) { // nothing to gain: the per-unit wavefront
            g->fplan = lf_block_plan();
            g->fplan_phase_block.clear();
        }
    }
    g->finalized = true;
    return LF_OK;
}

int64_t lf_dist_graph_num_noncontiguous(const lf_dist_graph *g) { return g ? g->n_noncontiguous : -1; }

int64_t lf_dist_graph_state_size(const lf_dist_graph *g)
{
    return g ? g->N + (int64_t)(g->ghost_target[0].size() + g->ghost_target[1].size()) : -1;
}
int lf_dist_graph_num_phases(const lf_dist_graph *g) { return g ? g->nphases : -1; }
int64_t lf_dist_graph_num_launch_units(const lf_dist_graph *g) { return g ? (int64_t)g->level_start.size() - 1 : -1; }

int lf_dist_graph_get_layout(const lf_dist_graph *g, int32_t *perm, int32_t *phase_of_position)
{
    if (!g || !g->finalized) return lf_set_error(LF_E_INVALID, "graph not finalized");
    if (perm) std::memcpy(perm, g->perm.data(), sizeof(int32_t) * g->perm.size());
    if (phase_of_position)
        for (int64_t p = 0; p < g->N; ++p) phase_of_position[p] = g->phase[g->perm[p]];
    return LF_OK;
}

// plan getters for host-side executors / tests: CSR in position space
int lf_dist_graph_get_csr(const lf_dist_graph *g, int32_t *ups_ptr, int32_t *ups_idx, int64_t *n_edges)
{
    if (!g || !g->finalized) return lf_set_error(LF_E_INVALID, "graph not finalized");
    if (n_edges) *n_edges = (int64_t)g->ups_idx.size();
    if (ups_ptr) std::memcpy(ups_ptr, g->ups_ptr.data(), sizeof(int32_t) * g->ups_ptr.size());
    if (ups_idx) std::memcpy(ups_idx, g->ups_idx.data(), sizeof(int32_t) * g->ups_idx.size());
    return LF_OK;
}

// position range [begin, end) of the cells of `phase`
int lf_dist_graph_phase_range(const lf_dist_graph *g, int phase, int64_t out[2])
{
    if (!g || !g->finalized || phase < 0 || phase >= g->nphases) return lf_set_error(LF_E_INVALID, "bad argument");
    out[0] = g->level_start[g->phase_level[phase]];
    out[1] = g->level_start[g->phase_level[phase + 1]];
    return LF_OK;
}

// position range [begin, end) of one part of a phase (0 = boundary-critical, 1 = bulk)
int lf_dist_graph_part_range(const lf_dist_graph *g, int phase, int part, int64_t out[2])
{
    if (!g || !g->finalized || phase < 0 || phase >= g->nphases || part < 0 || part > 1) return lf_set_error(LF_E_INVALID, "bad argument");
    out[0] = g->level_start[g->stage_level[2 * phase + part]];
    out[1] = g->level_start[g->stage_level[2 * phase + part + 1]];
    return LF_OK;
}

// round j: out = {send_top, send_bottom, recv_top, recv_bottom} element counts
int lf_dist_graph_round_counts(const lf_dist_graph *g, int round, int64_t out[4])
{
    if (!g || !g->finalized || round < 0 || round >= g->nphases) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int side = 0; side < 2; ++side) {
        out[side] = g->export_off[side][round + 1] - g->export_off[side][round];
        out[2 + side] = g->ghost_off[side][round + 1] - g->ghost_off[side][round];
    }
    return LF_OK;
}

// positions (in the state vector) of the cells sent in `round` to `side`, and the first slot the values
// received from `side` land in
int lf_dist_graph_round_send_positions(const lf_dist_graph *g, int round, int side, int32_t *positions)
{
    if (!g || !g->finalized || round < 0 || round >= g->nphases || side < 0 || side > 1)
        return lf_set_error(LF_E_INVALID, "bad argument");
    const int64_t a = g->export_off[side][round], b = g->export_off[side][round + 1];
    for (int64_t i = a; i < b; ++i) positions[i - a] = g->export_pos[side][i];
    return LF_OK;
}
// fused sub-steps: out[0] = slab slots, [1..2] first export slot top / bottom, [3..4] first ghost slot top / bottom,
// [5] first slot of the local cells that feed a later phase
int lf_dist_graph_slab_layout(const lf_dist_graph *g, int64_t out[6])
{
    if (!g || !g->finalized || !out) return lf_set_error(LF_E_INVALID, "graph not finalized");
    out[0] = g->n_slots;
    out[1] = g->slot_export[0];
    out[2] = g->slot_export[1];
    out[3] = g->slot_ghost[0];
    out[4] = g->slot_ghost[1];
    out[5] = g->slot_xphase;
    return LF_OK;
}
// level blocks of the fused path: out = {blocks, blocks of more than one level, cones, entries of the cone table}
int lf_dist_graph_block_stats(const lf_dist_graph *g, int64_t out[4])
{
    if (!g || !g->finalized || !out) return lf_set_error(LF_E_INVALID, "graph not finalized");
    out[0] = out[1] = out[2] = out[3] = 0;
    if (g->fplan_phase_block.empty()) return LF_OK;
    const lf_block_plan &f = g->fplan;
    out[0] = (int64_t)f.level.size() - 1;
    for (size_t b = 0; b + 1 < f.level.size(); ++b) {
        out[1] += f.level[b + 1] - f.level[b] > 1;
        out[2] += f.row[b + 1] - f.row[b] - 1;
    }
    out[3] = (int64_t)f.cone.size();
    return LF_OK;
}
// the block plan of single router calls (k_sweep_cones<DIST>), for tests: sizes = {stages + 1, blocks + 1, rows (= blocks
// + 1 entries), entries of the cone table, launch units + 1}; with the arrays NULL only the sizes are returned.  All sizes 0
// when no block holds more than one unit.
int lf_dist_graph_get_route_plan(const lf_dist_graph *g, int64_t sizes[5], int32_t *stage_block, int32_t *level, int32_t *row,
                                 int32_t *off, int32_t *cone, int64_t *level_start)
{
    if (!g || !g->finalized || !sizes) return lf_set_error(LF_E_INVALID, "graph not finalized");
    sizes[0] = sizes[1] = sizes[2] = sizes[3] = 0;
    sizes[4] = (int64_t)g->level_start.size();
    if (level_start) std::memcpy(level_start, g->level_start.data(), sizeof(int64_t) * g->level_start.size());
    if (g->rplan_stage_block.empty()) return LF_OK;
    const lf_block_plan &f = g->rplan;
    sizes[0] = (int64_t)g->rplan_stage_block.size();
    sizes[1] = (int64_t)f.level.size();
    sizes[2] = (int64_t)f.row.size();
    sizes[3] = (int64_t)f.cone.size();
    if (stage_block) std::memcpy(stage_block, g->rplan_stage_block.data(), sizeof(int32_t) * g->rplan_stage_block.size());
    if (level) std::memcpy(level, f.level.data(), sizeof(int32_t) * f.level.size());
    if (row) std::memcpy(row, f.row.data(), sizeof(int32_t) * f.row.size());
    if (off) std::memcpy(off, f.off.data(), sizeof(int32_t) * f.off.size());
    if (cone) std::memcpy(cone, f.cone.data(), sizeof(int32_t) * f.cone.size());
    return LF_OK;
}
// the block plan of the fused sub-step path (one plan per phase, cones of <= 256 cells per unit), same layout and calling
// convention; sizes = {phases + 1, blocks + 1, rows, entries of the cone table}
int lf_dist_graph_get_fused_plan(const lf_dist_graph *g, int64_t sizes[4], int32_t *phase_block, int32_t *level, int32_t *row,
                                 int32_t *off, int32_t *cone)
{
    if (!g || !g->finalized || !sizes) return lf_set_error(LF_E_INVALID, "graph not finalized");
    sizes[0] = sizes[1] = sizes[2] = sizes[3] = 0;
    if (g->fplan_phase_block.empty()) return LF_OK;
    const lf_block_plan &f = g->fplan;
    sizes[0] = (int64_t)g->fplan_phase_block.size();
    sizes[1] = (int64_t)f.level.size();
    sizes[2] = (int64_t)f.row.size();
    sizes[3] = (int64_t)f.cone.size();
    if (phase_block) std::memcpy(phase_block, g->fplan_phase_block.data(), sizeof(int32_t) * g->fplan_phase_block.size());
    if (level) std::memcpy(level, f.level.data(), sizeof(int32_t) * f.level.size());
    if (row) std::memcpy(row, f.row.data(), sizeof(int32_t) * f.row.size());
    if (off) std::memcpy(off, f.off.data(), sizeof(int32_t) * f.off.size());
    if (cone) std::memcpy(cone, f.cone.data(), sizeof(int32_t) * f.cone.size());
    return LF_OK;
}
// the fused path's tables by position: out_slot[N], ups_idx_f[n_edges] (either may be NULL)
int lf_dist_graph_get_fused_tables(const lf_dist_graph *g, int32_t *out_slot, int32_t *ups_idx_f)
{
    if (!g || !g->finalized) return lf_set_error(LF_E_INVALID, "graph not finalized");
    if (out_slot) std::memcpy(out_slot, g->out_slot.data(), sizeof(int32_t) * g->out_slot.size());
    if (ups_idx_f) std::memcpy(ups_idx_f, g->ups_idx_f.data(), sizeof(int32_t) * g->ups_idx_f.size());
    return LF_OK;
}

int64_t lf_dist_graph_round_recv_slot(const lf_dist_graph *g, int round, int side)
{
    if (!g || !g->finalized || round < 0 || round >= g->nphases || side < 0 || side > 1) return -1;
    return g->N + g->ghost_base[side] + g->ghost_off[side][round];
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// RCCL communicator (librccl is loaded lazily so that single-GPU use never touches it)
// ------------------------------------------------------------------------------------------------
namespace {
typedef struct { char internal[128]; } rccl_unique_id;
typedef void *rccl_comm_t;
struct rccl_api {
    void *lib = nullptr;
    int (*GetUniqueId)(rccl_unique_id *) = nullptr;
    int (*CommInitRank)(rccl_comm_t *, int, rccl_unique_id, int) = nullptr;
    int (*CommDestroy)(rccl_comm_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, rccl_comm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
rccl_api g_rccl;
constexpr int kNcclFloat64 = 8; // ncclDouble (rccl.h ncclDataType_t)

int load_rccl()
{
    if (g_rccl.lib) return LF_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return lf_set_error(LF_E_COMM, "cannot load librccl: %s", dlerror());
#define LF_SYM(field, name)                                                              \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                          \
    if (!g_rccl.field) return lf_set_error(LF_E_COMM, "librccl lacks symbol %s", name)
    LF_SYM(GetUniqueId, "ncclGetUniqueId");
    LF_SYM(CommInitRank, "ncclCommInitRank");
    LF_SYM(CommDestroy, "ncclCommDestroy");
    LF_SYM(Send, "ncclSend");
    LF_SYM(Recv, "ncclRecv");
    LF_SYM(GroupStart, "ncclGroupStart");
    LF_SYM(GroupEnd, "ncclGroupEnd");
    LF_SYM(GetErrorString, "ncclGetErrorString");
#undef LF_SYM
    g_rccl.lib = h;
    return LF_OK;
}
#define LF_NCCL(call)                                                                                   \
    do {                                                                                                \
        int e_ = (call);                                                                                \
        if (e_ != 0) return lf_set_error(LF_E_COMM, "%s failed: %s", #call, g_rccl.GetErrorString(e_)); \
    } while (0)
} // namespace

struct lf_comm {
    rccl_comm_t comm = nullptr;
    int nranks = 1, rank = 0, device = 0;
};

extern "C" {

int lf_comm_unique_id(char id[128])
{
    if (!id) return lf_set_error(LF_E_INVALID, "null argument");
    LF_TRY(load_rccl());
    rccl_unique_id u;
    LF_NCCL(g_rccl.GetUniqueId(&u));
    std::memcpy(id, u.internal, 128);
    return LF_OK;
}

int lf_comm_create(const char id[128], int nranks, int rank, int device, lf_comm **out)
{
    if (!id || !out) return lf_set_error(LF_E_INVALID, "null argument");
    LF_TRY(load_rccl());
    LF_TRY(lf_ctx(device, nullptr));
    rccl_unique_id u;
    std::memcpy(u.internal, id, 128);
    lf_comm *c = new lf_comm();
    c->nranks = nranks;
    c->rank = rank;
    c->device = device;
    int e = g_rccl.CommInitRank(&c->comm, nranks, u, rank);
    if (e != 0) {
        delete c;
        return lf_set_error(LF_E_COMM, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(e));
    }
    *out = c;
    return LF_OK;
}

void lf_comm_destroy(lf_comm *c)
{
    if (!c) return;
    if (c->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->comm);
    delete c;
}

// lf_comm_destroy that reports what the communicator's teardown returns: an operation that failed on the device after the
// last call that could have reported it (RCCL: an asynchronous error) surfaces here instead of being dropped
int lf_comm_close(lf_comm *c)
{
    if (!c) return LF_OK;
    int e = 0;
    if (c->comm && g_rccl.CommDestroy) e = g_rccl.CommDestroy(c->comm);
    delete c;
    if (e != 0)
        return lf_set_error(LF_E_COMM, "ncclCommDestroy: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "error");
    return LF_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// device: distributed router
// ------------------------------------------------------------------------------------------------
namespace {
struct dsegment {
    int k0, k1;
    bool wide;
};

__global__ void __launch_bounds__(kBlock) k_pack2(int n0, int n1, const int *__restrict__ pos0, const int *__restrict__ pos1,
                                                  const double *__restrict__ q, double *__restrict__ buf0,
                                                  double *__restrict__ buf1)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n0)
        buf0[i] = q[pos0[i]];
    else if (i < n0 + n1)
        buf1[i - n0] = q[pos1[i - n0]];
}
__global__ void __launch_bounds__(kBlock) k_dgather(int n, const int *__restrict__ perm, const double *__restrict__ src,
                                                    double *__restrict__ dst)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) dst[p] = src[perm[p]];
}
__global__ void __launch_bounds__(kBlock) k_dscatter(int n, const int *__restrict__ perm, const double *__restrict__ src,
                                                     double *__restrict__ dst)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) dst[perm[p]] = src[p];
}
} // namespace

struct lf_dist_router {
    int device = 0;
    lf_device_ctx *ctx = nullptr;
    int64_t N = 0, state_size = 0;
    int nphases = 1, kmax = 8;
    double beta = 0, inv_beta = 0, b_minus_1 = 0, dx_scalar = 0, dt = 0;
    bool has_floodplains = false, dx_per_pixel = false, fused = false;
    lf_dbuf<unsigned int> derived_ok; // fused sub-steps: flags of k_check_derived (fused_args::recompute)
    lf_dbuf<int32_t> perm, ups_ptr, ups_idx, ups_base, export_pos[2];
    lf_dbuf<long long> level_start;
    lf_dbuf<double> a1, a2, dx, constant, sendbuf[2];
    lf_dbuf<lf_rec24> rec24_1, rec24_2; // the level kernel's static values as one record per cell and section (dist_level_statics)
    bool statics_refused = false;
    std::vector<int64_t> h_level_start;
    std::vector<std::vector<dsegment>> schedule; // per stage (2 * phase + part, see lf_dist_graph::crit)
    // halo exchange beside the bulk part of a phase: second stream + events (lf_dist_router_route)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_part0 = nullptr, ev_halo = nullptr;
    // pipelined calls (lf_dist_router_route_many): second state vector, events per (call parity, round)
    lf_dbuf<double> pp_state;
    std::vector<hipEvent_t> pp_crit, pp_halo; // [2 * nphases]
    std::vector<int64_t> export_off[2], ghost_off[2];
    int64_t ghost_base[2] = {0, 0};
    int64_t last_launches = 0;
    // fused sub-steps (see lf_dist_graph): fused upstream lists, slab slot of every cell, router-output parity buffers,
    // the slabs [slot][sub-step] of both sections, sized for slab_steps sub-steps
    lf_dbuf<int32_t> ups_idx_f, out_slot;
    lf_dbuf<double> fused_qr1, fused_qr2, slab1, slab2;
    int64_t n_slots = 0, slab_steps = 0, slot_export[2] = {0, 0}, slot_ghost[2] = {0, 0};
    lf_dbuf<double> fused_hist1, fused_hist2; // [nsteps][N] router outputs of every sub-step (k_fused_level_steps<DIST>)
    size_t fused_hist_refused = SIZE_MAX;     // smallest history size that did not fit its budget (lf_history_ensure)
    std::vector<int32_t> phase_level; // [nphases + 1] first launch unit of every phase
    // level blocks + cones of every phase (empty: one launch per unit)
    std::vector<int> fb_level, fb_row, fb_off;
    std::vector<int32_t> fb_phase_block;
    lf_dbuf<int> fb_level_dev, fb_row_dev, fb_off_dev, fb_cone;
    // level blocks + cones of every stage for single router calls (lf_dist_graph::rplan; empty: the segment schedule)
    std::vector<int> rb_level, rb_row, rb_off;
    std::vector<int32_t> rb_stage_block;
    lf_dbuf<int> rb_cone;
};

namespace {

// The static values a cell of the level kernel reads (a, dx, its upstream list range, ups_base) as ONE 24-byte record per
// section instead of four streams (k_level<.., STATICS = 3>, lf_sweep.h), built on the first call; nullptr: scalar dx,
// LF_LEVEL_STATICS=0, a graph of 2^28 cells or more, or no memory for them.
const lf_rec24 *dist_level_statics(lf_dist_router *r, const sweep_args &A)
{
    if (!r->fused || !r->dx_per_pixel || r->statics_refused || r->N <= 0 || r->N >= ((int64_t)1 << 28)) return nullptr;
    const char *e = std::getenv("LF_LEVEL_STATICS");
    if (e && e[0] == '0') return nullptr;
    lf_dbuf<lf_rec24> &buf = (A.a == r->a1.p) ? r->rec24_1 : r->rec24_2;
    if (!buf.p) {
        if (r->ups_idx.n >= ((size_t)1 << 28) || buf.alloc((size_t)r->N) != LF_OK) {
            r->statics_refused = true;
            (void)hipGetLastError();
            return nullptr;
        }
        hipLaunchKernelGGL(k_static_records_indexed, dim3((unsigned)((r->N + kLevelBlock - 1) / kLevelBlock)), dim3(kLevelBlock), 0,
                           r->ctx->stream, (long long)r->N, A.a, (const double *)r->dx.p, (const int *)r->ups_ptr.p,
                           (const int *)r->ups_base.p, buf.p);
    }
    return buf.p;
}

// part: 0 = the boundary-critical cells of the phase, 1 = the rest, -1 = both
int dist_compute_phase(lf_dist_router *r, double *q, const double *lat, int section, int phase, int part = -1,
                       const double *q_in = nullptr)
{
    if (section != LF_SECTION_MAIN && section != LF_SECTION_FLOODPLAINS)
        return lf_set_error(LF_E_SECTION, "The section parameter must be either 'main_channel' or 'floodplain'!");
    if (section == LF_SECTION_FLOODPLAINS && !r->has_floodplains)
        return lf_set_error(LF_E_SECTION, "floodplains routing requested but alpha_floodplains was not given");
    if (phase < 0 || phase >= r->nphases) return lf_set_error(LF_E_INVALID, "phase %d out of range", phase);
    hipStream_t s = r->ctx->stream;
    sweep_args A;
    A.ups_ptr = r->ups_ptr.p;
    A.ups_idx = r->ups_idx.p;
    A.ups_base = r->ups_base.p;
    A.perm = nullptr;
    A.a = section == LF_SECTION_MAIN ? r->a1.p : r->a2.p;
    A.constant = r->constant.p;
    A.lat = lat;
    A.dx = r->dx_per_pixel ? r->dx.p : nullptr;
    A.dx_scalar = r->dx_scalar;
    A.beta = r->beta;
    A.inv_beta = r->inv_beta;
    A.b_minus_1 = r->b_minus_1;
    A.kmax = r->kmax;
    A.qord = q;
    A.q_pix = nullptr;
    A.adx = nullptr;
    A.rec24 = dist_level_statics(r, A);
    A.qold_src = q_in; // pipelined calls: old discharge from the other state vector (beta = 3/5 path only)
    if (q_in && !r->fused) return lf_set_error(LF_E_INVALID, "separate input discharge needs the beta = 3/5 path");
    if (!r->fused && phase == 0 && part != 1 && r->N > 0) { // general beta: constant for ALL local cells from the old discharge
        const int n = (int)r->N;
        hipLaunchKernelGGL(k_prep, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, (const int *)nullptr, q, lat, A.a, A.dx,
                           r->dx_scalar, r->beta, r->constant.p);
        r->last_launches++;
    }
    const char *cones_env = std::getenv("LF_ROUTE_CONES"); // (read at every call, as the single-domain router does)
    const bool cones = !r->rb_stage_block.empty() && !(cones_env && cones_env[0] == '0');
    for (int st = 2 * phase + (part == 1 ? 1 : 0); st <= 2 * phase + (part == 0 ? 0 : 1); ++st) {
        if (cones) { // blocks of units cone by cone (k_sweep_cones<DIST>), single wide units by the level kernel
            for (int b = r->rb_stage_block[st]; b < r->rb_stage_block[st + 1]; ++b) {
                const int k0 = r->rb_level[b], nl = r->rb_level[b + 1] - k0;
                if (nl > 1) {
                    cone_plan_args C;
                    C.cone = r->rb_cone.p + r->rb_off[b];
                    C.nl = nl;
                    C.n_cells = 0; // (unused by k_sweep_cones_dist)
                    const dim3 grid((unsigned)(r->rb_row[b + 1] - r->rb_row[b] - 1)), block(64);
                    if (r->fused)
                        hipLaunchKernelGGL((k_sweep_cones_dist<true>), grid, block, 0, s, C, A);
                    else
                        hipLaunchKernelGGL((k_sweep_cones_dist<false>), grid, block, 0, s, C, A);
                } else {
                    const int first = (int)r->h_level_start[k0];
                    const int count = (int)(r->h_level_start[k0 + 1] - r->h_level_start[k0]);
                    if (count <= 0) continue;
                    const dim3 grid(level_blocks_for(count)), block(kLevelBlock);
                    if (r->fused && A.rec24)
                        hipLaunchKernelGGL((k_level<true, true, true, 3>), grid, block, 0, s, first, count, A);
                    else if (r->fused)
                        hipLaunchKernelGGL((k_level<true, true, true>), grid, block, 0, s, first, count, A);
                    else
                        hipLaunchKernelGGL((k_level<false, true, true>), grid, block, 0, s, first, count, A);
                }
                r->last_launches++;
            }
            continue;
        }
    for (const dsegment &g : r->schedule[st]) {
        if (g.wide) {
            const int first = (int)r->h_level_start[g.k0];
            const int count = (int)(r->h_level_start[g.k1] - r->h_level_start[g.k0]);
            const dim3 grid(level_blocks_for(count)), block(kLevelBlock);
            if (r->fused && A.rec24)
                hipLaunchKernelGGL((k_level<true, true, true, 3>), grid, block, 0, s, first, count, A);
            else if (r->fused)
                hipLaunchKernelGGL((k_level<true, true, true>), grid, block, 0, s, first, count, A);
            else
                hipLaunchKernelGGL((k_level<false, true, true>), grid, block, 0, s, first, count, A);
        } else {
            const dim3 grid(1), block(kNarrowBlock);
            if (r->fused)
                hipLaunchKernelGGL((k_levels_narrow<true, true, true>), grid, block, 0, s, g.k0, g.k1, r->level_start.p,
                                   A);
            else
                hipLaunchKernelGGL((k_levels_narrow<false, true, true>), grid, block, 0, s, g.k0, g.k1,
                                   r->level_start.p, A);
        }
        r->last_launches++;
    }
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int dist_pack(lf_dist_router *r, const double *q, int round, hipStream_t s = nullptr)
{
    if (!s) s = r->ctx->stream;
    // both neighbours' send buffers in ONE launch (a launch boundary of the tail phases is worth more than the kernel)
    const int64_t a0 = r->export_off[0][round], n0 = r->export_off[0][round + 1] - a0;
    const int64_t a1 = r->export_off[1][round], n1 = r->export_off[1][round + 1] - a1;
    if (n0 + n1 > 0) {
        hipLaunchKernelGGL(k_pack2, dim3(blocks_for(n0 + n1)), dim3(kBlock), 0, s, (int)n0, (int)n1,
                           r->export_pos[0].p + a0, r->export_pos[1].p + a1, q, r->sendbuf[0].p + a0,
                           r->sendbuf[1].p + a1);
        r->last_launches++;
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}

} // namespace

extern "C" {

int lf_dist_router_create(const lf_dist_graph *g, const double *alpha, double beta, const double *dx, double dx_scalar,
                          double dt, const double *alpha_floodplains, int device, lf_dist_router **out)
{
    if (!g || !alpha || !out) return lf_set_error(LF_E_INVALID, "null argument");
    if (!g->finalized) return lf_set_error(LF_E_INVALID, "graph not finalized");
    lf_device_ctx *ctx;
    LF_TRY(lf_ctx(device, &ctx));
    lf_dist_router *r = new lf_dist_router();
    r->device = device;
    r->ctx = ctx;
    r->N = g->N;
    r->state_size = lf_dist_graph_state_size(g);
    r->nphases = g->nphases;
    r->kmax = std::min(8, g->K);
    r->beta = beta;
    r->inv_beta = 1 / beta;
    r->b_minus_1 = beta - 1;
    r->dx_scalar = dx_scalar;
    r->dt = dt;
    r->dx_per_pixel = dx != nullptr;
    r->has_floodplains = alpha_floodplains != nullptr;
    const char *force_general = std::getenv("LF_GENERAL_POW");
    r->fused = (beta == 0.6) && !(force_general && force_general[0] == '1');
    const int64_t n = g->N;
    int rc = LF_OK;
    {
        std::vector<double> h(n);
        auto fill = [&](const double *al) {
            for (int64_t p = 0; p < n; ++p) {
                const int32_t pix = g->perm[p];
                h[p] = al[pix] * (dx ? dx[pix] : dx_scalar) / dt; // kinematic_wave_parallel.py:127
            }
        };
        fill(alpha);
        rc = r->a1.upload(h.data(), n);
        if (rc == LF_OK && alpha_floodplains) {
            fill(alpha_floodplains);
            rc = r->a2.upload(h.data(), n);
        }
        if (rc == LF_OK && dx) {
            for (int64_t p = 0; p < n; ++p) h[p] = dx[g->perm[p]];
            rc = r->dx.upload(h.data(), n);
        }
    }
    if (rc == LF_OK) rc = r->perm.upload(g->perm.data(), n);
    if (rc == LF_OK) rc = r->ups_ptr.upload(g->ups_ptr.data(), n + 1);
    if (rc == LF_OK) rc = r->ups_idx.upload(g->ups_idx.data(), g->ups_idx.size());
    if (rc == LF_OK) {
        const char *e = std::getenv("LF_DIST_ALL_INDEXED"); // A/B switch: every cell through the index list
        if (e && e[0] == '1') {
            std::vector<int32_t> none(g->ups_base.size(), -1);
            rc = r->ups_base.upload(none.data(), none.size());
        } else
            rc = r->ups_base.upload(g->ups_base.data(), g->ups_base.size());
    }
    if (rc == LF_OK) {
        std::vector<long long> ls(g->level_start.begin(), g->level_start.end());
        rc = r->level_start.upload(ls.data(), ls.size());
    }
    if (rc == LF_OK && !r->fused) rc = r->constant.alloc(n);
    for (int side = 0; side < 2 && rc == LF_OK; ++side) {
        rc = r->export_pos[side].upload(g->export_pos[side].data(), g->export_pos[side].size());
        if (rc == LF_OK) rc = r->sendbuf[side].alloc(g->export_pos[side].size());
        r->export_off[side] = g->export_off[side];
        r->ghost_off[side] = g->ghost_off[side];
        r->ghost_base[side] = g->ghost_base[side];
    }
    if (rc == LF_OK) rc = r->ups_idx_f.upload(g->ups_idx_f.data(), g->ups_idx_f.size());
    if (rc == LF_OK) rc = r->out_slot.upload(g->out_slot.data(), g->out_slot.size());
    if (rc != LF_OK) {
        delete r;
        return rc;
    }
    if (!g->fplan_phase_block.empty()) {
        rc = r->fb_level_dev.upload(g->fplan.level.data(), g->fplan.level.size());
        if (rc == LF_OK) rc = r->fb_row_dev.upload(g->fplan.row.data(), g->fplan.row.size());
        if (rc == LF_OK) rc = r->fb_off_dev.upload(g->fplan.off.data(), g->fplan.off.size());
        if (rc == LF_OK) rc = r->fb_cone.upload(g->fplan.cone.data(), g->fplan.cone.size());
        if (rc != LF_OK) {
            delete r;
            return rc;
        }
        r->fb_level = g->fplan.level;
        r->fb_row = g->fplan.row;
        r->fb_off = g->fplan.off;
        r->fb_phase_block = g->fplan_phase_block;
    }
    if (!g->rplan_stage_block.empty()) {
        const int rc2 = r->rb_cone.upload(g->rplan.cone.data(), g->rplan.cone.size());
        if (rc2 != LF_OK) {
            delete r;
            return rc2;
        }
        r->rb_level = g->rplan.level;
        r->rb_row = g->rplan.row;
        r->rb_off = g->rplan.off;
        r->rb_stage_block = g->rplan_stage_block;
    }
    r->n_slots = g->n_slots;
    for (int side = 0; side < 2; ++side) {
        r->slot_export[side] = g->slot_export[side];
        r->slot_ghost[side] = g->slot_ghost[side];
    }
    r->phase_level = g->phase_level;
    r->h_level_start = g->level_start;
    r->schedule.resize(2 * (size_t)g->nphases);
    for (int j = 0; j < 2 * g->nphases; ++j) {
        const int64_t k_end = g->stage_level[j + 1];
        for (int64_t k = g->stage_level[j]; k < k_end;) {
            const int64_t size = g->level_start[k + 1] - g->level_start[k];
            if (size > kNarrowMax) {
                r->schedule[j].push_back({(int)k, (int)k + 1, true});
                ++k;
            } else {
                int64_t e = k + 1;
                while (e < k_end && g->level_start[e + 1] - g->level_start[e] <= kNarrowMax) ++e;
                r->schedule[j].push_back({(int)k, (int)e, false});
                k = e;
            }
        }
    }
    *out = r;
    return LF_OK;
}

void lf_dist_router_destroy(lf_dist_router *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->ctx->stream);
    if (r->comm_stream) {
        (void)hipStreamSynchronize(r->comm_stream);
        (void)hipStreamDestroy(r->comm_stream);
    }
    if (r->ev_part0) (void)hipEventDestroy(r->ev_part0);
    if (r->ev_halo) (void)hipEventDestroy(r->ev_halo);
    for (hipEvent_t e : r->pp_crit) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->pp_halo) (void)hipEventDestroy(e);
    delete r;
}

int64_t lf_dist_router_state_size(const lf_dist_router *r) { return r ? r->state_size : -1; }
int64_t lf_dist_router_last_launches(const lf_dist_router *r) { return r ? r->last_launches : -1; }

// local pixel order -> engine order (first N entries of the state vector; ghost slots untouched) and back
int lf_dist_router_to_engine_order(lf_dist_router *r, const double *src_pix_dev, double *dst_ord_dev)
{
    if (!r || !src_pix_dev || !dst_ord_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    const int n = (int)r->N;
    if (n > 0)
        hipLaunchKernelGGL(k_dgather, dim3(blocks_for(n)), dim3(kBlock), 0, r->ctx->stream, n, r->perm.p, src_pix_dev,
                           dst_ord_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}
int lf_dist_router_from_engine_order(lf_dist_router *r, const double *src_ord_dev, double *dst_pix_dev)
{
    if (!r || !src_ord_dev || !dst_pix_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    const int n = (int)r->N;
    if (n > 0)
        hipLaunchKernelGGL(k_dscatter, dim3(blocks_for(n)), dim3(kBlock), 0, r->ctx->stream, n, r->perm.p, src_ord_dev,
                           dst_pix_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

// sweep of the local cells of one phase (state vector q_ord_dev has lf_dist_router_state_size entries)
int lf_dist_router_compute_phase(lf_dist_router *r, double *q_ord_dev, const double *lat_ord_dev, int section, int phase)
{
    if (!r || !q_ord_dev || !lat_ord_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    return dist_compute_phase(r, q_ord_dev, lat_ord_dev, section, phase);
}

// one part of a phase: 0 = the boundary-critical cells (the exports of the phase and what drains into them inside the
// phase), 1 = the rest; the two are independent of each other
int lf_dist_router_compute_part(lf_dist_router *r, double *q_ord_dev, const double *lat_ord_dev, int section, int phase,
                                int part)
{
    if (!r || !q_ord_dev || !lat_ord_dev || part < 0 || part > 1) return lf_set_error(LF_E_INVALID, "bad argument");
    LF_HIP(hipSetDevice(r->device));
    return dist_compute_phase(r, q_ord_dev, lat_ord_dev, section, phase, part);
}

// gathers the boundary cells of `round` into the two send buffers; returns their device addresses / counts
int lf_dist_router_pack(lf_dist_router *r, const double *q_ord_dev, int round, void *send_ptr[2], int64_t send_count[2])
{
    if (!r || !q_ord_dev || round < 0 || round >= r->nphases) return lf_set_error(LF_E_INVALID, "bad argument");
    LF_HIP(hipSetDevice(r->device));
    LF_TRY(dist_pack(r, q_ord_dev, round));
    for (int side = 0; side < 2; ++side) {
        if (send_ptr) send_ptr[side] = r->sendbuf[side].p + r->export_off[side][round];
        if (send_count) send_count[side] = r->export_off[side][round + 1] - r->export_off[side][round];
    }
    return LF_OK;
}

// where the values received in `round` from each side land inside the state vector
int lf_dist_router_recv_slots(const lf_dist_router *r, int round, int64_t slot[2], int64_t count[2])
{
    if (!r || round < 0 || round >= r->nphases) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int side = 0; side < 2; ++side) {
        slot[side] = r->N + r->ghost_base[side] + r->ghost_off[side][round];
        count[side] = r->ghost_off[side][round + 1] - r->ghost_off[side][round];
    }
    return LF_OK;
}

// pack + RCCL Send/Recv with the rank above (rank_top) and below (rank_bottom); -1 = no neighbour
static int dist_exchange(lf_dist_router *r, lf_comm *comm, double *q_ord_dev, int round, int rank_top, int rank_bottom,
                         hipStream_t s)
{
    if (!r || !comm || !q_ord_dev || round < 0 || round >= r->nphases) return lf_set_error(LF_E_INVALID, "bad argument");
    LF_HIP(hipSetDevice(r->device));
    LF_TRY(dist_pack(r, q_ord_dev, round, s));
    const int peer[2] = {rank_top, rank_bottom};
    bool any = false;
    for (int side = 0; side < 2; ++side) {
        const int64_t ns = r->export_off[side][round + 1] - r->export_off[side][round];
        const int64_t nr = r->ghost_off[side][round + 1] - r->ghost_off[side][round];
        if ((ns > 0 || nr > 0) && peer[side] < 0) return lf_set_error(LF_E_INVALID, "halo traffic without a neighbour rank");
        any = any || ns > 0 || nr > 0;
    }
    if (!any) return LF_OK;
    LF_NCCL(g_rccl.GroupStart());
    for (int side = 0; side < 2; ++side) {
        const int64_t ns = r->export_off[side][round + 1] - r->export_off[side][round];
        const int64_t nr = r->ghost_off[side][round + 1] - r->ghost_off[side][round];
        if (ns > 0)
            LF_NCCL(g_rccl.Send(r->sendbuf[side].p + r->export_off[side][round], (size_t)ns, kNcclFloat64, peer[side],
                                comm->comm, s));
        if (nr > 0)
            LF_NCCL(g_rccl.Recv(q_ord_dev + r->N + r->ghost_base[side] + r->ghost_off[side][round], (size_t)nr,
                                kNcclFloat64, peer[side], comm->comm, s));
    }
    LF_NCCL(g_rccl.GroupEnd());
    return LF_OK;
}

// The halo stream and the two events the overlapped route call orders it with, each created once whichever entry point
// comes first (a pipelined route_many before the first plain route used to leave the events null)
static int ensure_comm_stream(lf_dist_router *r)
{
    if (!r->comm_stream) LF_HIP(hipStreamCreateWithFlags(&r->comm_stream, hipStreamNonBlocking));
    if (!r->ev_part0) LF_HIP(hipEventCreateWithFlags(&r->ev_part0, hipEventDisableTiming));
    if (!r->ev_halo) LF_HIP(hipEventCreateWithFlags(&r->ev_halo, hipEventDisableTiming));
    return LF_OK;
}

int lf_dist_router_exchange(lf_dist_router *r, lf_comm *comm, double *q_ord_dev, int round, int rank_top, int rank_bottom)
{
    if (!r) return lf_set_error(LF_E_INVALID, "null argument");
    return dist_exchange(r, comm, q_ord_dev, round, rank_top, rank_bottom, r->ctx->stream);
}

// one kinematicWaveRouting call on the partitioned raster (asynchronous on the library stream)
int lf_dist_router_route(lf_dist_router *r, lf_comm *comm, double *q_ord_dev, const double *lat_ord_dev, int section,
                         int rank_top, int rank_bottom)
{
    if (!r || !q_ord_dev || !lat_ord_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    r->last_launches = 0;
    // Two streams (LF_DIST_OVERLAP=0: one): the boundary-critical part of phase j runs first, the round-j halo (pack +
    // RCCL Send/Recv) then goes to the communication stream and runs BESIDE the bulk part of the phase; phase j + 1 waits
    // for it.  The packs read export cells (part 0, final), the receives write ghost slots of round j that only later
    // phases read: no buffer is touched by both streams at once.
    const char *e = std::getenv("LF_DIST_OVERLAP");
    const bool overlap = comm && !(e && e[0] == '0');
    if (overlap) LF_TRY(ensure_comm_stream(r));
    hipStream_t s = r->ctx->stream;
    for (int j = 0; j < r->nphases; ++j) {
        bool any = false;
        if (j + 1 < r->nphases)
            for (int side = 0; side < 2; ++side)
                any = any || r->export_off[side][j + 1] > r->export_off[side][j] ||
                      r->ghost_off[side][j + 1] > r->ghost_off[side][j];
        if (any && !comm) return lf_set_error(LF_E_COMM, "halo exchange needed but no communicator given");
        if (!any) {
            LF_TRY(dist_compute_phase(r, q_ord_dev, lat_ord_dev, section, j));
        } else if (!overlap) {
            LF_TRY(dist_compute_phase(r, q_ord_dev, lat_ord_dev, section, j));
            LF_TRY(dist_exchange(r, comm, q_ord_dev, j, rank_top, rank_bottom, s));
        } else {
            LF_TRY(dist_compute_phase(r, q_ord_dev, lat_ord_dev, section, j, 0));
            LF_HIP(hipEventRecord(r->ev_part0, s));
            LF_HIP(hipStreamWaitEvent(r->comm_stream, r->ev_part0, 0));
            LF_TRY(dist_exchange(r, comm, q_ord_dev, j, rank_top, rank_bottom, r->comm_stream));
            LF_HIP(hipEventRecord(r->ev_halo, r->comm_stream));
            LF_TRY(dist_compute_phase(r, q_ord_dev, lat_ord_dev, section, j, 1));
            LF_HIP(hipStreamWaitEvent(s, r->ev_halo, 0));
        }
    }
    return LF_OK;
}

// one part of a phase with the old discharge read from q_in and everything else (new discharge, upstream values, ghost
// slots) in q_out: the building block of the pipelined calls, exposed for tests and other transports
int lf_dist_router_compute_part_io(lf_dist_router *r, const double *q_in_dev, double *q_out_dev, const double *lat_ord_dev,
                                   int section, int phase, int part)
{
    if (!r || !q_in_dev || !q_out_dev || !lat_ord_dev || part < 0 || part > 1) return lf_set_error(LF_E_INVALID, "bad argument");
    LF_HIP(hipSetDevice(r->device));
    return dist_compute_phase(r, q_out_dev, lat_ord_dev, section, phase, part, q_in_dev == q_out_dev ? nullptr : q_in_dev);
}

// ncalls kinematicWaveRouting calls in a row (lat_ord_dev[s] = lateral inflow of call s), software-pipelined: call s
// reads the old discharge from one state vector and writes the other (the caller's and a second one of the router,
// alternating), so phase 0 of call s + 1 -- nearly all of the work on a raster with short flow paths -- does not have to
// wait for the later phases of call s and runs beside their halo rounds.  Order on the compute stream per call s:
//     [part 0 of phase 0 of call s+1 -> round 0 of call s+1 to the communication stream]
//     part 0 of phase 1 of call s -> round 1 of call s to the communication stream
//     [bulk of phase 0 of call s+1]            <- hides both rounds
//     bulk of phase 1 of call s, phases 2.. of call s (each: wait for the round before, part 0, its round, bulk)
// Every kernel runs on the compute stream in this order; only the exchanges run beside them, and each touches ghost slots
// (or export cells) that no kernel between its issue and its wait reads (or writes).  The result -- identical to ncalls
// calls of lf_dist_router_route -- ends in q_ord_dev.  beta = 3/5 path with a communicator; otherwise call by call.
int lf_dist_router_route_many(lf_dist_router *r, lf_comm *comm, double *q_ord_dev, const double *const *lat_ord_dev, int ncalls,
                              int section, int rank_top, int rank_bottom)
{
    if (!r || !q_ord_dev || !lat_ord_dev || ncalls < 0) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int s = 0; s < ncalls; ++s)
        if (!lat_ord_dev[s]) return lf_set_error(LF_E_INVALID, "null argument");
    const char *e = std::getenv("LF_DIST_OVERLAP");
    const bool pipelined = comm && r->fused && r->nphases > 1 && ncalls > 1 && !(e && e[0] == '0');
    if (!pipelined) {
        for (int s = 0; s < ncalls; ++s)
            LF_TRY(lf_dist_router_route(r, comm, q_ord_dev, lat_ord_dev[s], section, rank_top, rank_bottom));
        return LF_OK;
    }
    LF_HIP(hipSetDevice(r->device));
    const int P = r->nphases;
    LF_TRY(ensure_comm_stream(r));
    if (r->pp_crit.empty()) {
        r->pp_crit.assign(2 * (size_t)P, nullptr);
        r->pp_halo.assign(2 * (size_t)P, nullptr);
        for (size_t i = 0; i < r->pp_crit.size(); ++i) {
            LF_HIP(hipEventCreateWithFlags(&r->pp_crit[i], hipEventDisableTiming));
            LF_HIP(hipEventCreateWithFlags(&r->pp_halo[i], hipEventDisableTiming));
        }
    }
    if (!r->pp_state.p) {
        LF_TRY(r->pp_state.alloc((size_t)std::max<int64_t>(r->state_size, 1)));
        LF_HIP(hipMemsetAsync(r->pp_state.p, 0, sizeof(double) * (size_t)std::max<int64_t>(r->state_size, 1), r->ctx->stream));
    }
    hipStream_t s0 = r->ctx->stream;
    double *B[2] = {q_ord_dev, r->pp_state.p};
    r->last_launches = 0;
    auto traffic = [&](int j) {
        if (j + 1 >= P) return false;
        for (int side = 0; side < 2; ++side)
            if (r->export_off[side][j + 1] > r->export_off[side][j] || r->ghost_off[side][j + 1] > r->ghost_off[side][j]) return true;
        return false;
    };
    // call c: in = B[c & 1], out = B[(c + 1) & 1]
    auto part = [&](int c, int j, int pt) {
        return dist_compute_phase(r, B[(c + 1) & 1], lat_ord_dev[c], section, j, pt, B[c & 1]);
    };
    auto issue_round = [&](int c, int j) -> int { // after part 0 of phase j of call c
        if (!traffic(j)) return LF_OK;
        hipEvent_t ec = r->pp_crit[(size_t)(c & 1) * P + j], eh = r->pp_halo[(size_t)(c & 1) * P + j];
        LF_HIP(hipEventRecord(ec, s0));
        LF_HIP(hipStreamWaitEvent(r->comm_stream, ec, 0));
        LF_TRY(dist_exchange(r, comm, B[(c + 1) & 1], j, rank_top, rank_bottom, r->comm_stream));
        LF_HIP(hipEventRecord(eh, r->comm_stream));
        return LF_OK;
    };
    auto wait_round = [&](int c, int j) -> int { // before phase j + 1 of call c
        if (!traffic(j)) return LF_OK;
        LF_HIP(hipStreamWaitEvent(s0, r->pp_halo[(size_t)(c & 1) * P + j], 0));
        return LF_OK;
    };
    LF_TRY(part(0, 0, 0));
    LF_TRY(issue_round(0, 0));
    LF_TRY(part(0, 0, 1));
    for (int c = 0; c < ncalls; ++c) {
        const bool next = c + 1 < ncalls;
        if (next) {
            LF_TRY(part(c + 1, 0, 0));
            LF_TRY(issue_round(c + 1, 0));
        }
        for (int j = 1; j < P; ++j) {
            LF_TRY(wait_round(c, j - 1));
            LF_TRY(part(c, j, 0));
            LF_TRY(issue_round(c, j));
            if (j == 1 && next) LF_TRY(part(c + 1, 0, 1));
            LF_TRY(part(c, j, 1));
        }
    }
    if (ncalls & 1) // the last call wrote the router's vector: hand the result back in the caller's
        LF_HIP(hipMemcpyAsync(q_ord_dev, r->pp_state.p, sizeof(double) * (size_t)r->N, hipMemcpyDeviceToDevice, s0));
    LF_HIP(hipGetLastError());
    return LF_OK;
}

// One routing.dynamic() sub-step (routing.py:512-603, 693-703) on the row-block partition: the element-wise stages run
// on the rank's own cells, each router call is lf_dist_router_route (sweeps + halo exchange per phase).  All vectors
// in the rank's engine order; ChanQKin and Chan2QKin are STATE vectors (lf_dist_router_state_size entries, ghost slots
// behind the N local cells), every other vector has N entries.  Equals lf_routing_substep on the whole raster.
int lf_dist_routing_substep(lf_dist_router *r, lf_comm *comm, const lf_substep_args *a, int rank_top, int rank_bottom)
{
    if (!r || !a) return lf_set_error(LF_E_INVALID, "null argument");
    if (!a->engine_order) return lf_set_error(LF_E_INVALID, "the partitioned sub-step needs engine-order vectors");
    if (a->split && !r->has_floodplains)
        return lf_set_error(LF_E_SECTION, "split routing requested but the router has no floodplain alpha");
    LF_TRY(lf_substep_stage(r->device, 0, r->N, a));
    LF_TRY(lf_dist_router_route(r, comm, a->ChanQKin, a->scratch0, LF_SECTION_MAIN, rank_top, rank_bottom));
    LF_TRY(lf_substep_stage(r->device, 1, r->N, a));
    if (a->split) {
        LF_TRY(lf_dist_router_route(r, comm, a->Chan2QKin, a->scratch1, LF_SECTION_FLOODPLAINS, rank_top, rank_bottom));
        LF_TRY(lf_substep_stage(r->device, 2, r->N, a));
    }
    return LF_OK;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------
// fused sub-steps on the partition: a whole model step (nsteps x routing.dynamic, routing.py:512-603) per phase
// ------------------------------------------------------------------------------------------------
namespace {

struct dist_fused_args {
    const int *__restrict__ ups_base; // [N] first upstream position if the upstream cells are consecutive (then: same phase)
    const int *__restrict__ ups_idx;  // fused lists: same-phase position, or -(slab slot) - 1
    const int *__restrict__ out_slot; // [N] slab slot of the cell's router outputs, -1 none
    int unit0, nunits;                // launch units (levels) of the phase
};

// k_fused_substeps (lf_fused.h) on one phase of a rank's block: launch t works on (unit t - s, sub-step s).  A cell
// reads the router outputs of its same-phase upstream cells from the parity buffers (previous unit, written at t - 1)
// and those of earlier phases / other ranks from the slabs; cells feeding a later phase or another rank store theirs in
// the slabs.  The per-cell arithmetic is fused_cell's: bit-identical to the single-domain wavefront.
template <bool SPLIT>
__global__ void __launch_bounds__(kBlock) k_fused_substeps_dist(fused_args F, dist_fused_args D)
{
    int s, blk;
    if (F.packed) {
        int cnt = 0, start = 0;
        for (int q = 0; q < F.nsteps; ++q) {
            const bool ge = (int)blockIdx.x >= F.blk_start[q];
            cnt += ge ? 1 : 0;
            start = ge ? F.blk_start[q] : start;
        }
        s = cnt - 1;
        blk = (int)blockIdx.x - start;
    } else {
        s = blockIdx.y;
        blk = blockIdx.x;
    }
    int k;
    if (F.use_lvl) { // beside k_fused_cones: the single (wide) level sub-step s works on at this wave time, -1 none
        k = F.lvl[s];
        if (k < 0) return;
    } else {
        k = F.t - s;
        if (k < 0 || k >= D.nunits) return;
        k += D.unit0;
    }
    const long long first = F.level_start[k];
    const long long i = (long long)blk * kBlock + threadIdx.x;
    if (i >= F.level_start[k + 1] - first) return;
    const long long p = first + i;
    const int u0 = F.ups_ptr[p], u1 = F.ups_ptr[p + 1];
    // ups_base also marks runs of GHOST slots (positions >= N of the per-call state vector) as consecutive: here those
    // come from the slabs, so only a run inside the local cells -- then: same phase, previous unit -- takes the short way
    const int base_raw = D.ups_base[p];
    const int base = (base_raw >= 0 && (long long)base_raw + (u1 - u0) <= F.n) ? base_raw : -1;
    const int kmax = F.kmax;
    const long long slot = D.out_slot[p];
    const int *idx = D.ups_idx;
    const double *slab1 = F.root1, *slab2 = F.root2;
    const long long ss = F.root_ss, off = (long long)s * F.root_st;
    fused_cell<SPLIT, false>(F, p, s, [=](const double *q, int section) {
        if (base >= 0) return upstream_sum8(q, base, base + (u1 - u0), kmax);
        const double *slab = section ? slab2 : slab1;
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double x = 0.0;
            if (j < kmax && u0 + j < u1) {
                const int e = idx[u0 + j];
                x = e >= 0 ? q[e] : slab[(long long)(-(e + 1)) * ss + off];
            }
            v[j] = x;
        }
        double ups = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) ups += v[j];
        return ups;
    }, slot);
}

int dist_fused_prepare(lf_dist_router *r, const lf_substep_args *a, int nsteps)
{
    if (!r || !a || nsteps < 1) return lf_set_error(LF_E_INVALID, "bad argument");
    if (!a->engine_order) return lf_set_error(LF_E_INVALID, "the partitioned sub-steps need engine-order vectors");
    if (a->split && !r->has_floodplains)
        return lf_set_error(LF_E_SECTION, "split routing requested but the router has no floodplain alpha");
    LF_HIP(hipSetDevice(r->device));
    const size_t n = (size_t)std::max<int64_t>(r->N, 1);
    if (!r->fused_qr1.p) LF_TRY(r->fused_qr1.alloc(2 * n));
    if (a->split && !r->fused_qr2.p) LF_TRY(r->fused_qr2.alloc(2 * n));
    const size_t need = (size_t)std::max<int64_t>(r->n_slots, 1) * (size_t)nsteps;
    if (r->slab_steps < nsteps || !r->slab1.p) {
        LF_HIP(hipStreamSynchronize(r->ctx->stream)); // nobody may still read the old slabs
        LF_TRY(r->slab1.alloc(need));
        r->slab2.release();
        r->slab_steps = nsteps;
    }
    if (a->split && !r->slab2.p) LF_TRY(r->slab2.alloc(need));
    // which derived statics the wavefront may recompute instead of streaming (fused_args::recompute)
    const char *e = std::getenv("LF_NO_RECOMPUTE");
    if (r->N > 0 && !(e && e[0] == '1')) {
        if (!r->derived_ok.p) LF_TRY(r->derived_ok.alloc(1));
        hipStream_t s = r->ctx->stream;
        LF_HIP(hipMemsetD32Async((hipDeviceptr_t)r->derived_ok.p, 3, 1, s));
        hipLaunchKernelGGL(k_check_derived, dim3(blocks_for(r->N)), dim3(kBlock), 0, s, (long long)r->N, *a, r->a1.p, r->a2.p,
                           r->dx_per_pixel ? r->dx.p : nullptr, r->dx_scalar, r->dt, r->derived_ok.p);
        LF_HIP(hipGetLastError());
    } else {
        r->derived_ok.release();
    }
    return LF_OK;
}

// every sub-step of one phase as a wavefront over (unit, sub-step): nunits + nsteps - 1 launches
int dist_fused_phase(lf_dist_router *r, const lf_substep_args *a, int nsteps, int64_t sideflow_stride, int phase, int msteps = 0,
                     int64_t side_mstride = 0)
{
    if (phase < 0 || phase >= r->nphases) return lf_set_error(LF_E_INVALID, "phase %d out of range", phase);
    if (sideflow_stride != 0 && sideflow_stride != r->N) return lf_set_error(LF_E_INVALID, "sideflow_stride must be 0 or N");
    if (nsteps != r->slab_steps && nsteps > r->slab_steps) return lf_set_error(LF_E_INVALID, "lf_dist_fused_prepare first");
    const int unit0 = r->phase_level[phase], nunits = r->phase_level[phase + 1] - unit0;
    if (nunits <= 0 || r->N == 0) return LF_OK;
    fused_args F;
    std::memset(&F, 0, sizeof(F));
    F.S = *a;
    F.ups_ptr = r->ups_ptr.p;
    F.a1 = r->a1.p;
    F.a2 = r->a2.p;
    F.dx = r->dx_per_pixel ? r->dx.p : nullptr;
    F.level_start = r->level_start.p;
    F.qr1 = r->fused_qr1.p;
    F.qr2 = r->fused_qr2.p;
    F.root1 = r->slab1.p;
    F.root2 = r->slab2.p;
    F.nroots = r->n_slots;
    F.root_ss = r->slab_steps; // [slot][sub-step]
    F.root_st = 1;
    F.n = r->N;
    F.side_stride = sideflow_stride;
    F.msteps = msteps > 0 ? msteps : nsteps; // several model steps per call: lf_dist_routing_model_steps_fused
    F.side_mstride = side_mstride;
    F.dx_scalar = r->dx_scalar;
    F.beta = r->beta;
    F.inv_beta = r->inv_beta;
    F.b_minus_1 = r->b_minus_1;
    F.kmax = r->kmax;
    F.nlevels = (int)r->h_level_start.size() - 1;
    F.nsteps = nsteps;
    F.solve35 = r->fused ? 1 : 0;
    F.recompute = r->derived_ok.p;
    F.dt = r->dt;
    dist_fused_args D;
    D.ups_base = r->ups_base.p;
    D.ups_idx = r->ups_idx_f.p;
    D.out_slot = r->out_slot.p;
    D.unit0 = unit0;
    D.nunits = nunits;
    hipStream_t s = r->ctx->stream;
    F.d_ups_base = D.ups_base;
    F.d_ups_idx = D.ups_idx;
    F.d_out_slot = D.out_slot;
    // ---- a phase of few, wide levels: level after level, every level through all its sub-steps (k_fused_level_steps<DIST>,
    // lf_fused.h; the single domain's rule and switches: LF_FUSED_TIME_MAJOR, LF_FUSED_TIME_MAJOR_LEVELS) -------------------
    {
        static const int tm_levels = [] {
            const char *e = std::getenv("LF_FUSED_TIME_MAJOR_LEVELS");
            return e ? std::atoi(e) : 192;
        }();
        const char *e = std::getenv("LF_FUSED_TIME_MAJOR");
        const int64_t cells = r->h_level_start[unit0 + nunits] - r->h_level_start[unit0];
        const bool want = e ? e[0] != '0' : (nunits <= tm_levels && cells >= 20000 * (int64_t)nunits);
        if (want && nsteps > 1) {
            const bool ok = lf_history_ensure(r->fused_hist1, r->fused_hist2, r->fused_hist_refused,
                                              (size_t)nsteps * (size_t)r->N, a->split);
            if (ok) {
                F.hist1 = r->fused_hist1.p;
                F.hist2 = r->fused_hist2.p;
                const bool all35 = r->fused && a->Beta == 0.6;
                for (int k = 0; k < nunits; ++k) {
                    const int64_t w = r->h_level_start[unit0 + k + 1] - r->h_level_start[unit0 + k];
                    if (w <= 0) continue;
                    const dim3 grid(blocks_for(w)), block(kBlock);
                    if (a->split && all35)
                        hipLaunchKernelGGL((k_fused_level_steps<true, true, true>), grid, block, 0, s, F, unit0 + k);
                    else if (a->split)
                        hipLaunchKernelGGL((k_fused_level_steps<true, false, true>), grid, block, 0, s, F, unit0 + k);
                    else if (all35)
                        hipLaunchKernelGGL((k_fused_level_steps<false, true, true>), grid, block, 0, s, F, unit0 + k);
                    else
                        hipLaunchKernelGGL((k_fused_level_steps<false, false, true>), grid, block, 0, s, F, unit0 + k);
                    r->last_launches++;
                }
                LF_HIP(hipGetLastError());
                return LF_OK;
            }
            // no room for the history inside its budget (remembered in fused_hist_refused): the skewed wavefront below
        }
    }
    if (!r->fb_phase_block.empty() && nsteps <= kMaxPackedSteps) { // blocks of levels, cone by cone (k_fused_cones<DIST>)
        const int b0 = r->fb_phase_block[phase], NB = r->fb_phase_block[phase + 1] - b0;
        F.fb_level = r->fb_level_dev.p;
        F.fb_row = r->fb_row_dev.p;
        F.fb_cone = r->fb_cone.p;
        F.fb_off = r->fb_off_dev.p;
        F.fb_block0 = b0;
        F.fb_nblocks = NB;
        auto cones = [&](int b) { return (int64_t)(r->fb_row[b + 1] - r->fb_row[b] - 1); };
        auto multi = [&](int b) { return r->fb_level[b + 1] - r->fb_level[b] > 1; };
        const bool all35 = r->fused && a->Beta == 0.6;
        for (int t = 0; t < NB + nsteps - 1; ++t) {
            F.t = t;
            int64_t acc = 0;
            for (int q = 0; q < nsteps; ++q) {
                F.blk_start[q] = (int)acc;
                const int b = t - q;
                if (b >= 0 && b < NB && multi(b0 + b)) acc += cones(b0 + b);
            }
            F.blk_start[nsteps] = (int)acc;
            if (acc >= ((int64_t)1 << 31)) return lf_set_error(LF_E_INVALID, "fused sub-steps: grid too large");
            if (acc > 0) {
                F.packed = 1;
                F.use_lvl = 0;
                const dim3 grid((unsigned)acc);
                if (a->split && all35)
                    hipLaunchKernelGGL((k_fused_cones<true, true, false, true>), grid, dim3(kBlock), 0, s, F);
                else if (a->split)
                    hipLaunchKernelGGL((k_fused_cones<true, false, false, true>), grid, dim3(kBlock), 0, s, F);
                else if (all35)
                    hipLaunchKernelGGL((k_fused_cones<false, true, false, true>), grid, dim3(kBlock), 0, s, F);
                else
                    hipLaunchKernelGGL((k_fused_cones<false, false, false, true>), grid, dim3(kBlock), 0, s, F);
                r->last_launches++;
            }
            int64_t acc1 = 0, widest = 0;
            for (int q = 0; q < nsteps; ++q) {
                F.blk_start[q] = (int)acc1;
                F.lvl[q] = -1;
                const int b = t - q;
                if (b >= 0 && b < NB && !multi(b0 + b)) {
                    const int k = r->fb_level[b0 + b];
                    const int64_t w = r->h_level_start[k + 1] - r->h_level_start[k];
                    F.lvl[q] = k;
                    acc1 += blocks_for(w);
                    widest = std::max(widest, w);
                }
            }
            F.blk_start[nsteps] = (int)acc1;
            if (acc1 > 0) {
                F.use_lvl = 1;
                F.packed = 0;
                dim3 grid(blocks_for(widest), nsteps);
                if (2 * acc1 <= (int64_t)blocks_for(widest) * nsteps && acc1 < ((int64_t)1 << 31)) {
                    F.packed = 1;
                    grid = dim3((unsigned)acc1, 1);
                }
                if (a->split)
                    hipLaunchKernelGGL((k_fused_substeps_dist<true>), grid, dim3(kBlock), 0, s, F, D);
                else
                    hipLaunchKernelGGL((k_fused_substeps_dist<false>), grid, dim3(kBlock), 0, s, F, D);
                r->last_launches++;
            }
        }
        LF_HIP(hipGetLastError());
        return LF_OK;
    }
    auto width = [&](int k) { return r->h_level_start[unit0 + k + 1] - r->h_level_start[unit0 + k]; };
    for (int t = 0; t < nunits + nsteps - 1; ++t) {
        const int k_lo = std::max(0, t - nsteps + 1), k_hi = std::min(nunits - 1, t);
        int64_t widest = 0;
        for (int k = k_lo; k <= k_hi; ++k) widest = std::max(widest, width(k));
        F.t = t;
        dim3 grid(blocks_for(widest), nsteps);
        F.packed = 0;
        if (nsteps <= kMaxPackedSteps) {
            int64_t acc = 0;
            for (int q = 0; q < nsteps; ++q) {
                F.blk_start[q] = (int)acc;
                const int k = t - q;
                if (k >= 0 && k < nunits) acc += blocks_for(width(k));
            }
            F.blk_start[nsteps] = (int)acc;
            if (2 * acc <= (int64_t)blocks_for(widest) * nsteps && acc < ((int64_t)1 << 31)) {
                F.packed = 1;
                grid = dim3((unsigned)std::max<int64_t>(acc, 1), 1);
            }
        }
        if (a->split)
            hipLaunchKernelGGL((k_fused_substeps_dist<true>), grid, dim3(kBlock), 0, s, F, D);
        else
            hipLaunchKernelGGL((k_fused_substeps_dist<false>), grid, dim3(kBlock), 0, s, F, D);
        r->last_launches++;
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}

} // namespace

extern "C" {

// slabs and parity buffers for nsteps sub-steps (idempotent; lf_dist_routing_substeps_fused calls it itself)
int lf_dist_fused_prepare(lf_dist_router *r, const lf_substep_args *a, int nsteps)
{
    LF_TRY(dist_fused_prepare(r, a, nsteps));
    r->last_launches = 0;
    return LF_OK;
}

// all nsteps sub-steps of the local cells of one phase (after lf_dist_fused_prepare and the halo of the phase before)
int lf_dist_fused_phase(lf_dist_router *r, const lf_substep_args *a, int nsteps, int64_t sideflow_stride, int phase)
{
    if (!r || !a) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    return dist_fused_phase(r, a, nsteps, sideflow_stride, phase);
}

// halo of `round` in the slabs of `section`: out = {send offset, send count, recv offset, recv count} in doubles from
// lf_dist_fused_slab (count = cells x slab sub-steps: a round's values of one neighbour are one contiguous block)
int lf_dist_fused_halo_block(const lf_dist_router *r, int round, int side, int64_t out[4])
{
    if (!r || !out || round < 0 || round >= r->nphases || side < 0 || side > 1) return lf_set_error(LF_E_INVALID, "bad argument");
    const int64_t S = r->slab_steps;
    out[0] = (r->slot_export[side] + r->export_off[side][round]) * S;
    out[1] = (r->export_off[side][round + 1] - r->export_off[side][round]) * S;
    out[2] = (r->slot_ghost[side] + r->ghost_off[side][round]) * S;
    out[3] = (r->ghost_off[side][round + 1] - r->ghost_off[side][round]) * S;
    return LF_OK;
}
int lf_dist_fused_slab(const lf_dist_router *r, int section, void **slab_dev)
{
    if (!r || !slab_dev) return lf_set_error(LF_E_INVALID, "null argument");
    *slab_dev = section == LF_SECTION_MAIN ? (void *)r->slab1.p : (void *)r->slab2.p;
    return LF_OK;
}

// RCCL Send/Recv of the round's slab blocks with the rank above / below, both sections in one group
int lf_dist_fused_exchange(lf_dist_router *r, lf_comm *comm, int round, int split, int rank_top, int rank_bottom)
{
    if (!r || round < 0 || round >= r->nphases) return lf_set_error(LF_E_INVALID, "bad argument");
    LF_HIP(hipSetDevice(r->device));
    const int peer[2] = {rank_top, rank_bottom};
    int64_t blk[2][4];
    bool any = false;
    for (int side = 0; side < 2; ++side) {
        LF_TRY(lf_dist_fused_halo_block(r, round, side, blk[side]));
        if ((blk[side][1] > 0 || blk[side][3] > 0) && peer[side] < 0)
            return lf_set_error(LF_E_INVALID, "halo traffic without a neighbour rank");
        any = any || blk[side][1] > 0 || blk[side][3] > 0;
    }
    if (!any) return LF_OK;
    if (!comm) return lf_set_error(LF_E_COMM, "halo exchange needed but no communicator given");
    hipStream_t s = r->ctx->stream;
    LF_NCCL(g_rccl.GroupStart());
    for (int section = 0; section < (split ? 2 : 1); ++section) {
        double *slab = section == 0 ? r->slab1.p : r->slab2.p;
        for (int side = 0; side < 2; ++side) {
            if (blk[side][1] > 0)
                LF_NCCL(g_rccl.Send(slab + blk[side][0], (size_t)blk[side][1], kNcclFloat64, peer[side], comm->comm, s));
            if (blk[side][3] > 0)
                LF_NCCL(g_rccl.Recv(slab + blk[side][2], (size_t)blk[side][3], kNcclFloat64, peer[side], comm->comm, s));
        }
    }
    LF_NCCL(g_rccl.GroupEnd());
    return LF_OK;
}

// Several MODEL steps per call on the partition (lf_routing_model_steps_fused on the whole raster, bit for bit): every phase
// runs the sub-steps of ALL n_model_steps model steps as one wavefront and hands over the slabs of all of them in ONE halo
// block -- the pipeline fill of a phase and the exchange round are paid once per call instead of once per model step, which
// is what the partition of a deep network needs (successive calls in flight).  a->SideflowChanM3: one
// vector per model step, sideflow_model_stride elements apart (0: one for all); a->sumDisDay: [n_model_steps][N], zeroed by
// the caller.
static int check_model_steps(const lf_dist_router *r, int steps_per_model_step, int n_model_steps, int64_t stride)
{
    if (!r || steps_per_model_step < 1 || n_model_steps < 1) return lf_set_error(LF_E_INVALID, "bad argument");
    if (stride != 0 && stride < r->N) return lf_set_error(LF_E_INVALID, "sideflow_model_stride must be 0 or >= N");
    return LF_OK;
}

int lf_dist_fused_phase_model_steps(lf_dist_router *r, const lf_substep_args *a, int steps_per_model_step, int n_model_steps,
                                    int64_t sideflow_model_stride, int phase)
{
    if (!a) return lf_set_error(LF_E_INVALID, "null argument");
    LF_TRY(check_model_steps(r, steps_per_model_step, n_model_steps, sideflow_model_stride));
    LF_HIP(hipSetDevice(r->device));
    return dist_fused_phase(r, a, steps_per_model_step * n_model_steps, 0, phase, steps_per_model_step, sideflow_model_stride);
}

int lf_dist_routing_model_steps_fused(lf_dist_router *r, lf_comm *comm, const lf_substep_args *a, int steps_per_model_step,
                                      int n_model_steps, int64_t sideflow_model_stride, int rank_top, int rank_bottom)
{
    if (!a) return lf_set_error(LF_E_INVALID, "null argument");
    LF_TRY(check_model_steps(r, steps_per_model_step, n_model_steps, sideflow_model_stride));
    const int nsteps = steps_per_model_step * n_model_steps;
    LF_TRY(dist_fused_prepare(r, a, nsteps));
    r->last_launches = 0;
    for (int j = 0; j < r->nphases; ++j) {
        LF_TRY(dist_fused_phase(r, a, nsteps, 0, j, steps_per_model_step, sideflow_model_stride));
        if (j + 1 < r->nphases) LF_TRY(lf_dist_fused_exchange(r, comm, j, a->split, rank_top, rank_bottom));
    }
    return LF_OK;
}

// nsteps x routing.dynamic() on the partition (= lf_routing_substeps_fused on the whole raster, bit for bit): phase by
// phase, every sub-step of a phase as one wavefront, ONE halo exchange per phase and model step (instead of one per
// phase, router call and sub-step).  All vectors in the rank's engine order, N entries (ChanQKin / Chan2QKin may be the
// state vectors of lf_dist_routing_substep: their ghost slots are not used here).  a->SideflowChanM3: sideflow_stride =
// 0 (one vector for all sub-steps) or N.
int lf_dist_routing_substeps_fused(lf_dist_router *r, lf_comm *comm, const lf_substep_args *a, int nsteps,
                                   int64_t sideflow_stride, int rank_top, int rank_bottom)
{
    LF_TRY(dist_fused_prepare(r, a, nsteps));
    r->last_launches = 0;
    for (int j = 0; j < r->nphases; ++j) {
        LF_TRY(dist_fused_phase(r, a, nsteps, sideflow_stride, j));
        if (j + 1 < r->nphases) LF_TRY(lf_dist_fused_exchange(r, comm, j, a->split, rank_top, rank_bottom));
    }
    return LF_OK;
}

} // extern "C"
