// lf_blocks.h -- host side: level blocks and their cones for the LDS sweeps (k_sweep_cones, k_fused_cones), shared by the
// single-domain router (lf_router.hip) and the row-block partition (lf_dist.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <vector>

// Workgroups go to the eight XCDs round robin by their linear id.  Neighbouring cones of a block share the 128-byte lines
// their level segments begin and end in (a segment of 64 cells = 4 lines + on average one shared): with the cones of a
// (block, sub-step) dealt out in launch order the two halves of such a line are fetched into two L2s.  This gives the
// workgroups of one XCD CONSECUTIVE cones of the range [0, n) instead: i = position in launch order, linear_id = of this
// workgroup (so position 0 has linear id linear_id - i); a bijection of [0, n), positions beyond n map to themselves.
// Device code (k_fused_cones) and host (lf_xcd_contiguous_order: the tests look at the whole order).
__host__ __device__ inline int lf_xcd_contiguous(int i, int n, unsigned linear_id)
{
    if (i >= n) return i;
    const unsigned start = linear_id - (unsigned)i; // linear id of position 0
    const int cls = (int)(linear_id & 7u);
    int before = 0, r_mine = 0;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int r = (int)(((unsigned)c - start) & 7u); // first position of class c
        const int cnt = r < n ? (n - r + 7) >> 3 : 0;
        before += c < cls ? cnt : 0;
        r_mine = c == cls ? r : r_mine;
    }
    return before + ((i - r_mine) >> 3);
}

// Runs of consecutive levels [k_lo, k_hi) of at most `wide` cells are cut into blocks of up to lmax levels; a wider level
// is a block of its own.  A block is cut into CONES: chunks of its last level, as long as possible with no level of the
// cone wider than `max_cone` cells, each with the range of every earlier level that drains into it (`child(pos)` = first
// position of the level before whose downstream cell is at or behind `pos`; the ranges tile every level of the block).
// A block whose thinnest possible cone (one cell of the last level) is still too wide somewhere loses levels until it
// fits (one level always does).
//   level[b]      first level of block b (absolute), level[B] = k_hi
//   row[b]        first cone row of block b; block b has row[b+1] - row[b] - 1 cones and one closing row
//   off[b]        first entry of block b in `cone`; a row holds one start per level of the block
struct lf_block_plan {
    std::vector<int> level, row, off, cone;
    bool any_multi = false; // some block holds more than one level
};

template <class CHILD>
inline void lf_build_level_blocks(const std::vector<int64_t> &ls, int64_t k_lo, int64_t k_hi, int lmax, int64_t wide,
                                  int max_cone, CHILD child, lf_block_plan &out)
{
    auto width = [&](int64_t k) { return ls[k + 1] - ls[k]; };
    if (out.row.empty()) out.row.push_back(0);
    std::vector<int64_t> st((size_t)std::max(lmax, 1)), en((size_t)std::max(lmax, 1));
    // starts of the cone above last-level position `pos` of the block [k0, k0 + nl): st[nl-1] = pos, st[j] = first
    // position of level j draining at or behind st[j+1] (the end of level j when st[j+1] is the end of level j+1, the
    // start of level j when it is the start)
    auto chain = [&](int64_t k0, int nl, int64_t pos, std::vector<int64_t> &o) {
        o[nl - 1] = pos;
        for (int j = nl - 2; j >= 0; --j) {
            if (pos >= ls[k0 + j + 2])
                pos = ls[k0 + j + 1];
            else if (pos <= ls[k0 + j + 1])
                pos = ls[k0 + j];
            else
                pos = child(pos);
            o[j] = pos;
        }
    };
    for (int64_t k = k_lo; k < k_hi;) {
        int nl = 1;
        if (width(k) <= wide)
            while (k + nl < k_hi && nl < lmax && width(k + nl) <= wide) ++nl;
        std::vector<int> rows; // starts, nl per cone, then the closing row
        for (;; --nl) {        // shrink until every cone fits a workgroup
            rows.clear();
            const int64_t lo = ls[k + nl - 1], hi = ls[k + nl];
            bool fits = true;
            for (int64_t a = lo; a < hi && fits;) {
                chain(k, nl, a, st);
                auto ok = [&](int64_t e) { // cone [a, e) of the last level: every level's range <= max_cone?
                    chain(k, nl, e, en);
                    for (int j = 0; j < nl; ++j)
                        if (en[j] - st[j] > max_cone) return false;
                    return true;
                };
                int64_t e = std::min<int64_t>(a + max_cone, hi);
                if (!ok(e)) { // largest e in (a, a + max_cone) that fits: the widths grow with e
                    int64_t good = a, bad = e;
                    while (bad - good > 1) {
                        const int64_t mid = good + (bad - good) / 2;
                        if (ok(mid)) good = mid; else bad = mid;
                    }
                    e = good;
                }
                if (e == a) { // even one cell of the last level has too wide a cone
                    fits = false;
                    break;
                }
                for (int j = 0; j < nl; ++j) rows.push_back((int)st[j]);
                a = e;
            }
            if (fits || nl == 1) break;
        }
        for (int j = 0; j < nl; ++j) rows.push_back((int)ls[k + j + 1]); // closing row: the end of every level
        out.level.push_back((int)k);
        out.off.push_back((int)out.cone.size());
        out.row.push_back(out.row.back() + (int)(rows.size() / nl));
        out.cone.insert(out.cone.end(), rows.begin(), rows.end());
        out.any_multi = out.any_multi || nl > 1;
        k += nl;
    }
}
