// bvh.cpp — binned-SAH BVH2 builder (host).  See bvh.h for the node format.
#include "bvh.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <thread>

namespace {

constexpr int kBins = 16;
constexpr float kTraversalCost = 1.0f;
// SAH cost of one triangle test relative to one node visit; PRT_BVH_CI overrides it for tuning runs
static float kIntersectCost = 1.5f;

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int a = 0; a < 3; ++a) {
            mn[a] = FLT_MAX;
            mx[a] = -FLT_MAX;
        }
    }
    void grow(const float* lo, const float* hi) {
        for (int a = 0; a < 3; ++a) {
            mn[a] = std::min(mn[a], lo[a]);
            mx[a] = std::max(mx[a], hi[a]);
        }
    }
    void grow(const Box& b) { grow(b.mn, b.mx); }
    float half_area() const {
        const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        if (dx < 0.0f) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
};

struct BuildNode {
    Box box;
    int32_t left = -1, right = -1;  // build-node indices
    uint32_t first = 0, count = 0;
    uint32_t depth = 0;
};

struct Builder {
    const float* verts;
    uint32_t n;
    uint32_t max_leaf;
    std::vector<float> tmin, tmax, cen;  // per-triangle bounds and centroids (3 floats each)
    std::vector<uint32_t> order;
    std::vector<BuildNode> nodes;
    std::atomic<uint32_t> next_node{0};

    uint32_t alloc_pair() { return next_node.fetch_add(2); }

    // Decide and perform the split of node `ni`.  Returns false if it became a leaf.
    bool split(uint32_t ni, bool force_median) {
        BuildNode& nd = nodes[ni];
        const uint32_t first = nd.first, count = nd.count;
        if (count <= 1) return false;
        Box cb;
        cb.reset();
        for (uint32_t k = first; k < first + count; ++k) {
            const float* c = &cen[3 * (size_t)order[k]];
            cb.grow(c, c);
        }
        int best_axis = -1, best_bin = -1;
        float best_cost = FLT_MAX;
        const float node_area = nd.box.half_area();
        if (!force_median) {
            for (int a = 0; a < 3; ++a) {
                const float ext = cb.mx[a] - cb.mn[a];
                if (!(ext > 0.0f)) continue;
                Box bins[kBins];
                uint32_t cnt[kBins];
                for (int b = 0; b < kBins; ++b) {
                    bins[b].reset();
                    cnt[b] = 0;
                }
                const float scale = (float)kBins / ext;
                for (uint32_t k = first; k < first + count; ++k) {
                    const uint32_t t = order[k];
                    int b = (int)((cen[3 * (size_t)t + a] - cb.mn[a]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    bins[b].grow(&tmin[3 * (size_t)t], &tmax[3 * (size_t)t]);
                    ++cnt[b];
                }
                float right_area[kBins];
                uint32_t right_cnt[kBins];
                Box acc;
                acc.reset();
                uint32_t c = 0;
                for (int b = kBins - 1; b > 0; --b) {
                    acc.grow(bins[b]);
                    c += cnt[b];
                    right_area[b] = acc.half_area();
                    right_cnt[b] = c;
                }
                acc.reset();
                c = 0;
                for (int b = 0; b < kBins - 1; ++b) {
                    acc.grow(bins[b]);
                    c += cnt[b];
                    if (c == 0 || right_cnt[b + 1] == 0) continue;
                    const float cost = acc.half_area() * (float)c + right_area[b + 1] * (float)right_cnt[b + 1];
                    if (cost < best_cost) {
                        best_cost = cost;
                        best_axis = a;
                        best_bin = b;
                    }
                }
            }
        }
        if (best_axis >= 0 && count <= max_leaf && node_area > 0.0f) {
            const float split_cost = kTraversalCost + kIntersectCost * best_cost / node_area;
            const float leaf_cost = kIntersectCost * (float)count;
            if (leaf_cost <= split_cost) return false;
        }
        uint32_t mid = 0;
        if (best_axis >= 0) {
            const int a = best_axis;
            const float scale = (float)kBins / (cb.mx[a] - cb.mn[a]);
            const float cmn = cb.mn[a];
            const int bb = best_bin;
            auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                int b = (int)((cen[3 * (size_t)t + a] - cmn) * scale);
                b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                return b <= bb;
            });
            mid = (uint32_t)(it - (order.begin() + first));
        }
        if (mid == 0 || mid == count) {
            if (count <= max_leaf) return false;
            // object-median split along the widest centroid axis (also the forced path for deep trees)
            int a = 0;
            float ext = cb.mx[0] - cb.mn[0];
            for (int k = 1; k < 3; ++k)
                if (cb.mx[k] - cb.mn[k] > ext) {
                    ext = cb.mx[k] - cb.mn[k];
                    a = k;
                }
            mid = count / 2;
            std::nth_element(order.begin() + first, order.begin() + first + mid, order.begin() + first + count,
                             [&](uint32_t x, uint32_t y) {
                                 const float cx = cen[3 * (size_t)x + a], cy = cen[3 * (size_t)y + a];
                                 return cx < cy || (cx == cy && x < y);
                             });
        }
        const uint32_t l = alloc_pair();
        BuildNode& L = nodes[l];
        BuildNode& R = nodes[l + 1];
        L.first = first;
        L.count = mid;
        R.first = first + mid;
        R.count = count - mid;
        L.depth = R.depth = nd.depth + 1;
        L.box.reset();
        R.box.reset();
        for (uint32_t k = L.first; k < L.first + L.count; ++k)
            L.box.grow(&tmin[3 * (size_t)order[k]], &tmax[3 * (size_t)order[k]]);
        for (uint32_t k = R.first; k < R.first + R.count; ++k)
            R.box.grow(&tmin[3 * (size_t)order[k]], &tmax[3 * (size_t)order[k]]);
        nd.left = (int32_t)l;
        nd.right = (int32_t)l + 1;
        return true;
    }

    void build_subtree(uint32_t root, uint32_t median_depth) {
        std::vector<uint32_t> stack;
        stack.push_back(root);
        while (!stack.empty()) {
            const uint32_t ni = stack.back();
            stack.pop_back();
            if (split(ni, nodes[ni].depth >= median_depth)) {
                stack.push_back((uint32_t)nodes[ni].right);
                stack.push_back((uint32_t)nodes[ni].left);
            }
        }
    }
};

}  // namespace

bool bvh_build(const float* verts, uint32_t n_tris, uint32_t max_leaf_size, int n_threads, uint32_t max_allowed_depth,
               BvhBuild* out) {
    out->nodes.clear();
    out->nodes4.clear();
    out->max_stack4 = 0;
    out->nodes8.clear();
    out->depth8 = 0;
    out->order.clear();
    out->max_depth = 0;
    out->max_leaf = 0;
    out->sah_cost = 0.0f;
    if (n_tris == 0) return true;
    if (const char* e = getenv("PRT_BVH_CI")) kIntersectCost = (float)atof(e);
    if (const char* e = getenv("PRT_BVH_MAXLEAF")) max_leaf_size = (uint32_t)atoi(e);
    if (max_leaf_size < 1) max_leaf_size = 1;
    if (max_leaf_size > 15) max_leaf_size = 15;
    if (n_threads <= 0) n_threads = (int)std::max(1u, std::thread::hardware_concurrency());
    n_threads = std::min(n_threads, 32);

    Builder b;
    b.verts = verts;
    b.n = n_tris;
    b.max_leaf = max_leaf_size;
    b.tmin.resize(3 * (size_t)n_tris);
    b.tmax.resize(3 * (size_t)n_tris);
    b.cen.resize(3 * (size_t)n_tris);
    b.order.resize(n_tris);
    b.nodes.resize(2 * (size_t)n_tris + 2);
    BuildNode& root = b.nodes[0];
    b.next_node = 2;  // slot 1 unused: children are allocated in pairs
    root.box.reset();
    for (uint32_t t = 0; t < n_tris; ++t) {
        const float* v = verts + 9 * (size_t)t;
        for (int a = 0; a < 3; ++a) {
            const float lo = std::min(v[a], std::min(v[3 + a], v[6 + a]));
            const float hi = std::max(v[a], std::max(v[3 + a], v[6 + a]));
            b.tmin[3 * (size_t)t + a] = lo;
            b.tmax[3 * (size_t)t + a] = hi;
            b.cen[3 * (size_t)t + a] = 0.5f * lo + 0.5f * hi;
        }
        b.order[t] = t;
        root.box.grow(&b.tmin[3 * (size_t)t], &b.tmax[3 * (size_t)t]);
    }
    root.first = 0;
    root.count = n_tris;
    root.depth = 0;

    // SAH until this depth, then forced object-median splits: bounds the depth by
    // median_depth + ceil(log2(n)) so the traversal stack (64 entries) can never overflow.
    const uint32_t median_depth = 40;

    // Phase 1 (serial): split the top of the tree until there are enough independent subtrees.
    std::vector<uint32_t> frontier{0};
    const size_t want_tasks = (size_t)n_threads * 8;
    const uint32_t min_task = 4096;
    while (n_threads > 1 && frontier.size() < want_tasks) {
        size_t big = frontier.size();
        uint32_t big_count = min_task;
        for (size_t i = 0; i < frontier.size(); ++i)
            if (b.nodes[frontier[i]].count > big_count) {
                big_count = b.nodes[frontier[i]].count;
                big = i;
            }
        if (big == frontier.size()) break;
        const uint32_t ni = frontier[big];
        frontier.erase(frontier.begin() + (long)big);
        if (b.split(ni, b.nodes[ni].depth >= median_depth)) {
            frontier.push_back((uint32_t)b.nodes[ni].left);
            frontier.push_back((uint32_t)b.nodes[ni].right);
        }
    }
    // Phase 2 (parallel): each remaining subtree works on its own disjoint slice of order[].
    if (n_threads > 1 && frontier.size() > 1) {
        std::atomic<size_t> next{0};
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&]() {
                while (true) {
                    const size_t i = next.fetch_add(1);
                    if (i >= frontier.size()) break;
                    b.build_subtree(frontier[i], median_depth);
                }
            });
        for (auto& t : th) t.join();
    } else {
        for (uint32_t ni : frontier) b.build_subtree(ni, median_depth);
    }

    // Flatten: internal nodes renumbered in depth-first preorder; triangle slots are order[] as is.
    struct Item {
        uint32_t build, dev;
    };
    auto is_leaf = [&](uint32_t ni) { return b.nodes[ni].left < 0; };
    if (n_tris >= (1u << 26)) return false;  // 26-bit triangle slots (traversal work items carry 6 lane bits)

    // ---- BVH8Q: collapse to 8 children, place them in octant-ordered slots, quantize (layout: bvh.h).  This also
    // defines the triangle slot order of ALL trees: the leaf children of one wide node become contiguous. ----
    std::vector<uint32_t> new_first(b.nodes.size(), 0);
    std::vector<uint32_t> order_new(n_tris);
    bool have8 = true;
    {
        std::vector<uint32_t> st{0};
        while (!st.empty()) {
            const uint32_t ni = st.back();
            st.pop_back();
            if (is_leaf(ni)) {
                if (b.nodes[ni].count > 3) have8 = false;
                continue;
            }
            st.push_back((uint32_t)b.nodes[ni].right);
            st.push_back((uint32_t)b.nodes[ni].left);
        }
    }
    if (have8) {
        struct Item8 {
            uint32_t build, dev, level;
        };
        // Which binary nodes become 8-wide nodes: the SAH-optimal collapse for this binary topology by dynamic
        // programming (after Ylitie et al. 2017, section 3): cost[n][i] = least total area of wide nodes needed to
        // represent the subtree of n with at most i+1 child slots of its wide parent.  The leaves are fixed, so the
        // triangle term is a constant and only wide-node visits (probability ~ area) are minimised.  A greedy
        // "expand the largest child" collapse leaves ~45 % of the nodes with two children; this one does not.
        const uint32_t n_build = b.next_node.load();
        std::vector<float> cost(8 * (size_t)n_build, 0.0f);   // [n][i], i = 0..6 <-> 1..7 slots; [n][7]: as a wide node's 8 slots
        std::vector<uint8_t> pick(8 * (size_t)n_build, 0);    // split k (slots given to the left child); 0 = "use i-1 slots"
        for (uint32_t ni = n_build; ni-- > 0;) {  // children have larger indices than their parent
            if (is_leaf(ni)) continue;                          // leaves cost nothing extra in any number of slots
            const float* cl = &cost[8 * (size_t)b.nodes[ni].left];
            const float* cr = &cost[8 * (size_t)b.nodes[ni].right];
            float* cn = &cost[8 * (size_t)ni];
            uint8_t* pn = &pick[8 * (size_t)ni];
            float dist[9];
            uint8_t dk[9];
            for (int j = 2; j <= 8; ++j) {  // distribute j slots over the two children
                float bestc = FLT_MAX;
                int bk = 1;
                for (int k = 1; k < j; ++k) {
                    const float c = cl[std::min(k, 7) - 1] + cr[std::min(j - k, 7) - 1];
                    if (c < bestc) {
                        bestc = c;
                        bk = k;
                    }
                }
                dist[j] = bestc;
                dk[j] = (uint8_t)bk;
            }
            cn[0] = dist[8] + b.nodes[ni].box.half_area();  // one slot: n is a wide node of its own
            pn[0] = 0;
            pn[7] = dk[8];
            cn[7] = dist[8];
            for (int i = 2; i <= 7; ++i) {
                if (dist[i] < cn[i - 2]) {
                    cn[i - 1] = dist[i];
                    pn[i - 1] = dk[i];
                } else {
                    cn[i - 1] = cn[i - 2];
                    pn[i - 1] = 0;
                }
            }
        }
        // children of the wide node rooted at binary node ni
        auto collapse8 = [&](uint32_t ni, uint32_t* kids) -> int {
            int n = 0;
            if (is_leaf(ni)) {
                kids[n++] = ni;
                return n;
            }
            struct Want {
                uint32_t node;
                int slots;
            };
            Want stk[32];
            int top = 0;
            const int k8 = pick[8 * (size_t)ni + 7];
            stk[top++] = {(uint32_t)b.nodes[ni].right, 8 - k8};
            stk[top++] = {(uint32_t)b.nodes[ni].left, k8};
            while (top > 0) {
                Want wn = stk[--top];
                if (is_leaf(wn.node)) {
                    kids[n++] = wn.node;
                    continue;
                }
                int i = std::min(wn.slots, 7);
                while (i > 1 && pick[8 * (size_t)wn.node + (i - 1)] == 0) --i;  // "use fewer slots"
                if (i == 1) {
                    kids[n++] = wn.node;  // stays an internal child (a wide node of its own)
                    continue;
                }
                const int k = pick[8 * (size_t)wn.node + (i - 1)];
                stk[top++] = {(uint32_t)b.nodes[wn.node].right, i - k};
                stk[top++] = {(uint32_t)b.nodes[wn.node].left, k};
            }
            return n;
        };
        std::deque<Item8> queue;  // breadth first: the top of the tree is contiguous in memory
        out->nodes8.assign(20, 0u);
        queue.push_back({0, 0, 1});
        uint32_t tri_cursor = 0;
        while (!queue.empty() && have8) {
            const Item8 it = queue.front();
            queue.pop_front();
            out->depth8 = std::max(out->depth8, it.level);
            uint32_t kids[8];
            const int n = collapse8(it.build, kids);
            const Box& nb = b.nodes[it.build].box;
            // slot assignment: slot s is visited FIRST by rays whose direction is negative exactly on the axes
            // whose bit is set in s, so it should hold the child lying farthest towards +axis on those axes and
            // towards -axis on the others; greedy on cost = dot(child centre - node centre, that diagonal)
            int slot_of[8], child_in[8];
            for (int k = 0; k < 8; ++k) slot_of[k] = child_in[k] = -1;
            float cost[8][8];
            for (int c = 0; c < n; ++c) {
                const Box& cb = b.nodes[kids[c]].box;
                float d[3];
                for (int a = 0; a < 3; ++a) d[a] = (0.5f * cb.mn[a] + 0.5f * cb.mx[a]) - (0.5f * nb.mn[a] + 0.5f * nb.mx[a]);
                for (int sl = 0; sl < 8; ++sl)
                    cost[c][sl] = (((sl >> 0) & 1) ? d[0] : -d[0]) + (((sl >> 1) & 1) ? d[1] : -d[1]) + (((sl >> 2) & 1) ? d[2] : -d[2]);
            }
            for (int round = 0; round < n; ++round) {
                int bc = -1, bs = -1;
                float best = -FLT_MAX;
                for (int c = 0; c < n; ++c) {
                    if (slot_of[c] >= 0) continue;
                    for (int sl = 0; sl < 8; ++sl)
                        if (child_in[sl] < 0 && (bc < 0 || cost[c][sl] > best)) {
                            best = cost[c][sl];
                            bc = c;
                            bs = sl;
                        }
                }
                slot_of[bc] = bs;
                child_in[bs] = bc;
            }
            // quantization grid per axis: origin = the node's own minimum, cell = 2^(eb - 127), 8 bits per plane
            uint32_t eb[3];
            double cell[3];
            for (int a = 0; a < 3; ++a) {
                const double p = nb.mn[a], ext = (double)nb.mx[a] - p;
                const double big = std::max(std::fabs((double)nb.mn[a]), std::fabs((double)nb.mx[a]));
                int e = ext > 0.0 ? (int)std::ceil(std::log2(ext / 255.0)) : -126;
                // keep p + q * cell exactly representable in a double (the containment checks below rely on it)
                if (big > 0.0) e = std::max(e, (int)std::floor(std::log2(big)) - 30);
                e = std::max(e, -126);
                while (std::ceil(ext / std::ldexp(1.0, e)) > 255.0) ++e;
                if (e > 126) have8 = false;
                eb[a] = (uint32_t)(e + 127);
                cell[a] = std::ldexp(1.0, e);
            }
            if (!have8) break;
            uint32_t* w = &out->nodes8[20 * (size_t)it.dev];
            uint32_t imask = 0, meta[8] = {0, 0, 0, 0, 0, 0, 0, 0}, qlo[3][8], qhi[3][8];
            const uint32_t child_base = (uint32_t)(out->nodes8.size() / 20);
            const uint32_t tri_base = tri_cursor;
            uint32_t n_inner = 0;
            for (int sl = 0; sl < 8; ++sl) {
                for (int a = 0; a < 3; ++a) {  // empty slot: inverted box, never referenced (meta = 0)
                    qlo[a][sl] = 255;
                    qhi[a][sl] = 0;
                }
                const int c = child_in[sl];
                if (c < 0) continue;
                const uint32_t kid = kids[c];
                const Box& cb = b.nodes[kid].box;
                for (int a = 0; a < 3; ++a) {
                    const double p = nb.mn[a];
                    double lo = std::floor(((double)cb.mn[a] - p) / cell[a]);
                    while (lo > 0.0 && p + lo * cell[a] > (double)cb.mn[a]) lo -= 1.0;
                    if (lo < 0.0) lo = 0.0;
                    double hi = std::ceil(((double)cb.mx[a] - p) / cell[a]);
                    while (p + hi * cell[a] < (double)cb.mx[a]) hi += 1.0;
                    if (hi > 255.0 || lo > hi) have8 = false;
                    qlo[a][sl] = (uint32_t)lo;
                    qhi[a][sl] = (uint32_t)hi;
                }
                if (is_leaf(kid)) {
                    const BuildNode& lf = b.nodes[kid];
                    const uint32_t off = tri_cursor - tri_base;
                    meta[sl] = (((1u << lf.count) - 1u) << 5) | off;  // unary count {1, 3, 7}; off <= 21
                    new_first[kid] = tri_cursor;
                    for (uint32_t k = 0; k < lf.count; ++k) order_new[tri_cursor + k] = b.order[lf.first + k];
                    tri_cursor += lf.count;
                } else {
                    imask |= 1u << sl;
                    meta[sl] = (1u << 5) | (24u + (uint32_t)sl);
                    ++n_inner;
                }
            }
            if (!have8) break;
            // the internal children get consecutive node indices in slot order
            out->nodes8.resize(out->nodes8.size() + 20 * (size_t)n_inner, 0u);
            w = &out->nodes8[20 * (size_t)it.dev];
            uint32_t rank = 0;
            for (int sl = 0; sl < 8; ++sl)
                if (imask & (1u << sl)) queue.push_back({kids[child_in[sl]], child_base + rank++, it.level + 1});
            const float px = nb.mn[0], py = nb.mn[1], pz = nb.mn[2];
            memcpy(&w[0], &px, 4);
            memcpy(&w[1], &py, 4);
            memcpy(&w[2], &pz, 4);
            w[3] = eb[0] | (eb[1] << 8) | (eb[2] << 16) | (imask << 24);
            w[4] = child_base;
            w[5] = tri_base;
            auto pack4 = [](const uint32_t* v) { return v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24); };
            w[6] = pack4(meta);
            w[7] = pack4(meta + 4);
            w[8] = pack4(qlo[0]);  w[9] = pack4(qlo[0] + 4);
            w[10] = pack4(qlo[1]); w[11] = pack4(qlo[1] + 4);
            w[12] = pack4(qlo[2]); w[13] = pack4(qlo[2] + 4);
            w[14] = pack4(qhi[0]); w[15] = pack4(qhi[0] + 4);
            w[16] = pack4(qhi[1]); w[17] = pack4(qhi[1] + 4);
            w[18] = pack4(qhi[2]); w[19] = pack4(qhi[2] + 4);
        }
        if (have8 && tri_cursor != n_tris) have8 = false;
    }
    if (!have8) {  // keep the builder's own order
        out->nodes8.clear();
        out->depth8 = 0;
        std::vector<uint32_t> st{0};
        while (!st.empty()) {
            const uint32_t ni = st.back();
            st.pop_back();
            if (is_leaf(ni)) {
                new_first[ni] = b.nodes[ni].first;
                continue;
            }
            st.push_back((uint32_t)b.nodes[ni].right);
            st.push_back((uint32_t)b.nodes[ni].left);
        }
        order_new = b.order;
    }
    auto leaf_ref = [&](uint32_t ni) -> int32_t {
        const BuildNode& nd = b.nodes[ni];
        return (int32_t) ~((new_first[ni] << 4) | nd.count);
    };
    uint32_t n_internal = 0;
    {
        std::vector<uint32_t> st{0};
        while (!st.empty()) {
            uint32_t ni = st.back();
            st.pop_back();
            if (is_leaf(ni)) continue;
            ++n_internal;
            st.push_back((uint32_t)b.nodes[ni].right);
            st.push_back((uint32_t)b.nodes[ni].left);
        }
    }
    double sah = 0.0;
    const float root_area = std::max(root.box.half_area(), 1e-30f);
    if (is_leaf(0)) {
        // single-leaf tree: one device node whose left child is the leaf and whose right child is empty
        out->nodes.assign(16, 0.0f);
        float* q = out->nodes.data();
        const Box& bx = root.box;
        q[0] = bx.mn[0]; q[1] = bx.mn[1]; q[2] = bx.mn[2]; q[3] = bx.mx[0]; q[4] = bx.mx[1]; q[5] = bx.mx[2];
        // right box: degenerate point far away
        for (int k = 6; k < 12; ++k) q[k] = 3.0e38f;
        int32_t l = leaf_ref(0), r = ~0;
        memcpy(&q[12], &l, 4);
        memcpy(&q[13], &r, 4);
        out->max_depth = 1;
        out->max_leaf = root.count;
        sah = kIntersectCost * root.count;
    } else {
        out->nodes.assign(16 * (size_t)n_internal, 0.0f);
        std::vector<Item> st;
        st.push_back({0, 0});
        uint32_t next_dev = 1;
        while (!st.empty()) {
            const Item it = st.back();
            st.pop_back();
            const BuildNode& nd = b.nodes[it.build];
            const uint32_t li = (uint32_t)nd.left, ri = (uint32_t)nd.right;
            const Box& lb = b.nodes[li].box;
            const Box& rb = b.nodes[ri].box;
            float* q = &out->nodes[16 * (size_t)it.dev];
            q[0] = lb.mn[0]; q[1] = lb.mn[1]; q[2] = lb.mn[2]; q[3] = lb.mx[0]; q[4] = lb.mx[1]; q[5] = lb.mx[2];
            q[6] = rb.mn[0]; q[7] = rb.mn[1]; q[8] = rb.mn[2]; q[9] = rb.mx[0]; q[10] = rb.mx[1]; q[11] = rb.mx[2];
            int32_t lref, rref;
            // preorder: the left subtree's internal nodes directly follow this node
            uint32_t ldev = 0, rdev = 0;
            if (is_leaf(li)) {
                lref = leaf_ref(li);
            } else {
                ldev = next_dev++;
                lref = (int32_t)ldev;
            }
            if (is_leaf(ri)) {
                rref = leaf_ref(ri);
            } else {
                rdev = next_dev++;
                rref = (int32_t)rdev;
            }
            memcpy(&q[12], &lref, 4);
            memcpy(&q[13], &rref, 4);
            sah += kTraversalCost * nd.box.half_area() / root_area;
            for (uint32_t ci : {li, ri}) {
                const BuildNode& c = b.nodes[ci];
                if (is_leaf(ci)) {
                    out->max_depth = std::max(out->max_depth, c.depth + 1);
                    out->max_leaf = std::max(out->max_leaf, c.count);
                    sah += kIntersectCost * c.count * c.box.half_area() / root_area;
                }
            }
            if (!is_leaf(ri)) st.push_back({ri, rdev});
            if (!is_leaf(li)) st.push_back({li, ldev});
        }
    }
    out->sah_cost = (float)sah;

    // ---- BVH4: collapse the binary tree (the child with the largest area is replaced by its two children
    // until a node has four children or only leaves are left).  128 B per node, see bvh.h. ----
    {
        struct Item4 {
            uint32_t build;  // build node whose (collapsed) children form this wide node
            uint32_t dev;
        };
        const float kInf = std::numeric_limits<float>::infinity();
        out->nodes4.clear();
        auto collapse = [&](uint32_t ni, uint32_t* kids) -> int {
            int n = 0;
            if (is_leaf(ni)) {
                kids[n++] = ni;
                return n;
            }
            kids[n++] = (uint32_t)b.nodes[ni].left;
            kids[n++] = (uint32_t)b.nodes[ni].right;
            while (n < 4) {
                int pick = -1;
                float best = -1.0f;
                for (int k = 0; k < n; ++k)
                    if (!is_leaf(kids[k])) {
                        const float a = b.nodes[kids[k]].box.half_area();
                        if (a > best) {
                            best = a;
                            pick = k;
                        }
                    }
                if (pick < 0) break;
                const uint32_t c = kids[pick];
                kids[pick] = (uint32_t)b.nodes[c].left;
                kids[n++] = (uint32_t)b.nodes[c].right;
            }
            return n;
        };
        std::vector<Item4> st;
        std::vector<uint32_t> need;  // per wide node: worst-case pushes below it (filled bottom-up afterwards)
        std::vector<std::array<int32_t, 4>> refs;
        st.push_back({0, 0});
        out->nodes4.assign(32, 0.0f);
        refs.push_back({-1, -1, -1, -1});
        while (!st.empty()) {
            const Item4 it = st.back();
            st.pop_back();
            uint32_t kids[4];
            const int n = collapse(it.build, kids);
            float* q = &out->nodes4[32 * (size_t)it.dev];
            std::array<int32_t, 4> r{-1, -1, -1, -1};
            for (int c = 0; c < 4; ++c) {
                if (c < n) {
                    const Box& bx = b.nodes[kids[c]].box;
                    q[0 + c] = bx.mn[0]; q[4 + c] = bx.mx[0];
                    q[8 + c] = bx.mn[1]; q[12 + c] = bx.mx[1];
                    q[16 + c] = bx.mn[2]; q[20 + c] = bx.mx[2];
                    if (is_leaf(kids[c])) {
                        r[c] = leaf_ref(kids[c]);
                    } else {
                        const uint32_t dev = (uint32_t)(out->nodes4.size() / 32);
                        out->nodes4.resize(out->nodes4.size() + 32, 0.0f);
                        q = &out->nodes4[32 * (size_t)it.dev];  // resize may have moved the storage
                        refs.push_back({-1, -1, -1, -1});
                        r[c] = (int32_t)dev;
                        st.push_back({kids[c], dev});
                    }
                } else {  // empty child: a box no ray can enter, empty leaf
                    q[0 + c] = q[4 + c] = q[8 + c] = q[12 + c] = q[16 + c] = q[20 + c] = kInf;
                    r[c] = ~0;
                }
            }
            memcpy(&q[24], r.data(), 16);
            refs[it.dev] = r;
        }
        // worst-case stack need: a node with m children pushes at most m-1 refs, then one child is entered
        const size_t n4 = refs.size();
        need.assign(n4, 0);
        for (size_t i = n4; i-- > 0;) {  // children always have larger indices than their parent
            uint32_t m = 0, deepest = 0;
            for (int c = 0; c < 4; ++c) {
                if (refs[i][c] == ~0) continue;
                ++m;
                if (refs[i][c] >= 0) deepest = std::max(deepest, need[(size_t)refs[i][c]]);
            }
            need[i] = (m ? m - 1 : 0) + deepest;
        }
        out->max_stack4 = need[0];
    }
    out->order = std::move(order_new);
    return out->max_depth <= max_allowed_depth;
}
