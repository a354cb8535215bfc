// bvh_build.h - host-side builder of the threaded bounding-volume hierarchy (device_params.h,
// "BVH").  Included by ptgpu_api.hip only.
//
// The reference scans every object for every ray (geomfunc.h:126-193).  For scenes whose cost is
// that scan over one large mesh (obj-model.txt, large-object.txt, complex.txt) the kernels may walk
// a hierarchy instead; a primitive is skipped only when the ray misses a padded box around it or
// enters that box beyond the closest hit found so far, and the closest hit is chosen by
// (distance, object index), so the result does not depend on the visiting order: it is the
// reference's.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <vector>

#include "pt_records.h"

namespace ptbvh {

constexpr int kLeaf = 4; // primitives per leaf, padded with never-hit records

struct Box {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    void grow(const double p[3]) {
        for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], p[a]);
            hi[a] = std::max(hi[a], p[a]);
        }
    }
    void grow(const Box &b) {
        grow(b.lo);
        grow(b.hi);
    }
    double area() const {
        const double x = hi[0] - lo[0], y = hi[1] - lo[1], z = hi[2] - lo[2];
        return x < 0 ? 0 : 2 * (x * y + y * z + z * x);
    }
};

struct Built {
    int type = -1;                   // PT_SPHERE or PT_TRIANGLE: the primitive type inside the hierarchy
    std::vector<float4> nodes;       // 2 quads per node: (lo.xyz, skip index) (hi.xyz, leaf: first primitive | inner: -1)
    std::vector<float4> prims;       // 3 quads per primitive slot, leaf order, kLeaf slots per leaf
    std::vector<int> prim_of_slot;   // slot -> primitive index (index order), -1 for a filler
    std::vector<int> rest;           // primitives of the other type (and non-finite ones): scanned one by one
    double sah = 0;                  // surface-area cost of the tree: sum over nodes of area / root area x (1 per inner
                                     // node test, kLeaf per leaf) - what a random ray that hits the root is expected to pay
};

inline int float_bits(float f) {
    int v;
    memcpy(&v, &f, sizeof v);
    return v;
}
inline float bits_float(int v) {
    float f;
    memcpy(&f, &v, sizeof f);
    return f;
}

// prims: 3 quads per primitive in index order (device_params.h); types[i]: PT_SPHERE / PT_TRIANGLE.
// keys (optional, with force_type): leaves never mix primitives of different keys and the leaf
// order runs key by key (keys ascending, -1 last) - used to lay the hierarchy over the shadow
// order, whose triangle groups are cluster-pure.  In that mode every primitive of the tree's type
// is in the tree (a non-finite one gets an all-space box).
inline Built build(const std::vector<float4> &prims, const std::vector<int> &types,
                   const std::vector<int> *keys = nullptr, int force_type = -1, bool all_axes = false) {
    // all_axes: the SAH bins are tried along all three centroid axes instead of the widest one only.  Neither is
    // better everywhere (measured: the 5 050-triangle bunny +25 %, obj-model.txt's teapot -8 %): the caller builds
    // both and keeps the tree with the lower cost (Built::sah).
    Built out;
    const size_t n = types.size();
    size_t n_sph = 0;
    for (int t : types) n_sph += t == PT_SPHERE;
    out.type = force_type >= 0 ? force_type : (n_sph > n - n_sph ? PT_SPHERE : PT_TRIANGLE);
    auto key_of = [&](int i) { return keys ? ((*keys)[i] < 0 ? 0x7fffffff : (*keys)[i]) : 0; };

    std::vector<Box> boxes(n);
    std::vector<int> order;
    for (size_t i = 0; i < n; ++i) {
        Box b;
        bool finite = true;
        if (types[i] == PT_SPHERE) {
            const float4 c = prims[3 * i];
            const double r = std::fabs((double)c.w);
            const double lo[3] = {c.x - r, c.y - r, c.z - r}, hi[3] = {c.x + r, c.y + r, c.z + r};
            b.grow(lo);
            b.grow(hi);
        } else {
            const float4 a = prims[3 * i], e1 = prims[3 * i + 1], e2 = prims[3 * i + 2];
            const double v[3][3] = {{a.x, a.y, a.z},
                                    {(double)a.x + e1.x, (double)a.y + e1.y, (double)a.z + e1.z},
                                    {(double)a.x + e2.x, (double)a.y + e2.y, (double)a.z + e2.z}};
            for (auto &p : v) b.grow(p);
        }
        for (int q = 0; q < 3; ++q) { // min/max drop a NaN, so look at the record itself
            const float4 v = prims[3 * i + q];
            finite = finite && std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z) && (q > 0 || std::isfinite(v.w));
        }
        for (int a = 0; a < 3; ++a) finite = finite && std::isfinite(b.lo[a]) && std::isfinite(b.hi[a]);
        // pad far beyond the rounding of the primitive tests: 1e-4 of the extent and position + 10 EPSILON
        for (int a = 0; a < 3; ++a) {
            const double pad = 1e-4 * ((b.hi[a] - b.lo[a]) + std::fabs(b.lo[a]) + std::fabs(b.hi[a])) + 10 * (double)PT_EPSILON;
            b.lo[a] -= pad;
            b.hi[a] += pad;
        }
        if (!finite && keys)
            for (int a = 0; a < 3; ++a) { b.lo[a] = -3e38; b.hi[a] = 3e38; }
        boxes[i] = b;
        if (types[i] == out.type && (finite || keys)) order.push_back((int)i);
        else out.rest.push_back((int)i);
    }
    if (keys) std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return key_of(x) < key_of(y); });

    struct Task { int begin, end; };
    // recursive median-of-SAH-bins build over order[begin, end), emitting nodes in preorder
    std::vector<int> leaf_first; // per node: first slot or -1
    std::vector<Box> node_box;
    std::vector<int> skip;
    std::function<void(int, int)> rec = [&](int begin, int end) {
        const int me = (int)node_box.size();
        Box nb, cb;
        for (int k = begin; k < end; ++k) {
            nb.grow(boxes[order[k]]);
            const Box &b = boxes[order[k]];
            const double c[3] = {(b.lo[0] + b.hi[0]) / 2, (b.lo[1] + b.hi[1]) / 2, (b.lo[2] + b.hi[2]) / 2};
            cb.grow(c);
        }
        node_box.push_back(nb);
        leaf_first.push_back(-1);
        skip.push_back(0);
        if (key_of(order[begin]) != key_of(order[end - 1])) { // split on the key boundary nearest the middle
            int best_mid = -1;
            for (int k = begin + 1; k < end; ++k)
                if (key_of(order[k - 1]) != key_of(order[k]) &&
                    (best_mid < 0 || std::abs(k - (begin + end) / 2) < std::abs(best_mid - (begin + end) / 2)))
                    best_mid = k;
            rec(begin, best_mid);
            rec(best_mid, end);
            skip[me] = (int)node_box.size();
            return;
        }
        if (end - begin <= kLeaf) {
            leaf_first[me] = (int)out.prim_of_slot.size();
            for (int k = 0; k < kLeaf; ++k) out.prim_of_slot.push_back(k < end - begin ? order[begin + k] : -1);
            skip[me] = (int)node_box.size();
            return;
        }
        // binned SAH over the three centroid axes (the widest one decides ties and degenerate ranges)
        int axis = 0;
        for (int a = 1; a < 3; ++a)
            if (cb.hi[a] - cb.lo[a] > cb.hi[axis] - cb.lo[axis]) axis = a;
        int mid = (begin + end) / 2;
        constexpr int kBins = 16; // (32 bins measured no better on any scene)
        double best = 1e300;
        int best_split = -1, best_axis = axis;
        for (int pick = 0; pick < (all_axes ? 3 : 1); ++pick) {
            const int a = (axis + pick) % 3;
            const double width = cb.hi[a] - cb.lo[a];
            if (!(width > 0)) continue;
            Box bin_box[kBins];
            int bin_n[kBins] = {};
            auto bin_of = [&](int i) { return std::min(kBins - 1, (int)(((boxes[i].lo[a] + boxes[i].hi[a]) / 2 - cb.lo[a]) / width * kBins)); };
            for (int k = begin; k < end; ++k) {
                bin_box[bin_of(order[k])].grow(boxes[order[k]]);
                bin_n[bin_of(order[k])]++;
            }
            for (int s = 1; s < kBins; ++s) {
                Box l, r;
                int nl = 0, nr = 0;
                for (int b = 0; b < s; ++b) if (bin_n[b]) { l.grow(bin_box[b]); nl += bin_n[b]; }
                for (int b = s; b < kBins; ++b) if (bin_n[b]) { r.grow(bin_box[b]); nr += bin_n[b]; }
                if (!nl || !nr) continue;
                // leaves hold kLeaf slots: count whole leaves
                const double cost = l.area() * ((nl + kLeaf - 1) / kLeaf) + r.area() * ((nr + kLeaf - 1) / kLeaf);
                if (cost < best) { best = cost; best_split = s; best_axis = a; }
            }
        }
        axis = best_axis;
        auto centre = [&](int i) { return (boxes[i].lo[axis] + boxes[i].hi[axis]) / 2; };
        if (best_split > 0) {
            const double width = cb.hi[axis] - cb.lo[axis];
            auto bin_of = [&](int i) { return std::min(kBins - 1, (int)((centre(i) - cb.lo[axis]) / width * kBins)); };
            auto it = std::stable_partition(order.begin() + begin, order.begin() + end,
                                            [&](int i) { return bin_of(i) < best_split; });
            mid = (int)(it - order.begin());
        }
        // Order the range along the axis and move the split to a whole number of leaves on the left,
        // so that at most one leaf per subtree is partly filled (a filler slot costs a full test).
        std::stable_sort(order.begin() + begin, order.begin() + end, [&](int x, int y) { return centre(x) < centre(y); });
        if (mid <= begin || mid >= end) mid = (begin + end) / 2; // all centroids equal: split by count
        {
            int left = ((mid - begin + kLeaf / 2) / kLeaf) * kLeaf;
            if (left <= 0) left = kLeaf;
            if (left >= end - begin) left = ((end - begin - 1) / kLeaf) * kLeaf;
            mid = begin + left;
        }
        rec(begin, mid);
        rec(mid, end);
        skip[me] = (int)node_box.size();
    };
    if (!order.empty()) rec(0, (int)order.size());

    const float ninf = -__builtin_inff();
    for (int id : out.prim_of_slot) {
        if (id >= 0) {
            for (int q = 0; q < 3; ++q) out.prims.push_back(prims[3 * id + q]);
        } else { // never-hit fillers, as in the shadow order
            out.prims.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
            out.prims.push_back(out.type == PT_SPHERE ? make_float4(ninf, 0.f, 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f));
            out.prims.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
        }
    }
    if (!node_box.empty() && node_box[0].area() > 0)
        for (size_t i = 0; i < node_box.size(); ++i)
            out.sah += node_box[i].area() / node_box[0].area() * (leaf_first[i] >= 0 ? (double)kLeaf : 1.0);
    auto down = [](double v) { return std::nextafter((float)v, -INFINITY) ; };
    auto up = [](double v) { return std::nextafter((float)v, INFINITY); };
    for (size_t i = 0; i < node_box.size(); ++i) {
        const Box &b = node_box[i];
        out.nodes.push_back(make_float4(down(b.lo[0]), down(b.lo[1]), down(b.lo[2]), bits_float(skip[i])));
        out.nodes.push_back(make_float4(up(b.hi[0]), up(b.hi[1]), up(b.hi[2]), bits_float(leaf_first[i])));
    }
    return out;
}

} // namespace ptbvh
