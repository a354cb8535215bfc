// Host-only checks of csrc/bvh_build.h (compiled and run by tests/test_bvh_build.py; no GPU).
// Invariants: every primitive of the tree's type sits in exactly one leaf slot; a node's box
// contains its primitives' (unpadded) extents; skip indices describe a proper preorder layout;
// a keyed build never mixes keys in a leaf, lays leaves out key by key and fills them.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "bvh_build.h"

static int failures = 0;
#define CHECK(cond, ...)                                                     \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__);        \
            std::fprintf(stderr, __VA_ARGS__);                               \
            std::fprintf(stderr, "\n");                                      \
            ++failures;                                                      \
        }                                                                    \
    } while (0)

struct Scene {
    std::vector<float4> prims;
    std::vector<int> types, keys;
};

static Scene random_scene(unsigned seed, int n_tri, int n_sph, int n_keys) {
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> pos(-50.f, 50.f), ext(-3.f, 3.f);
    Scene s;
    for (int i = 0; i < n_tri + n_sph; ++i) {
        const bool sphere = i % (n_tri + n_sph) >= n_tri;
        if (sphere) {
            const float r = 1.f + std::fabs(ext(rng));
            s.prims.push_back(make_float4(pos(rng), pos(rng), pos(rng), r));
            s.prims.push_back(make_float4(r * r, 0, 0, 0));
            s.prims.push_back(make_float4(0, 0, 0, 0));
            s.types.push_back(PT_SPHERE);
            s.keys.push_back(-1);
        } else {
            const int key = n_keys ? (int)(rng() % (n_keys + 1)) - 1 : -1; // -1 = no cluster
            const float cx = key >= 0 ? 20.f * key - 40.f : pos(rng);
            s.prims.push_back(make_float4(cx + ext(rng), pos(rng), pos(rng), 0));
            s.prims.push_back(make_float4(ext(rng), ext(rng), ext(rng), 0));
            s.prims.push_back(make_float4(ext(rng), ext(rng), ext(rng), 0));
            s.types.push_back(PT_TRIANGLE);
            s.keys.push_back(key);
        }
    }
    return s;
}

static void extent(const Scene &s, int i, double lo[3], double hi[3]) {
    const float4 a = s.prims[3 * i], b = s.prims[3 * i + 1], c = s.prims[3 * i + 2];
    if (s.types[i] == PT_SPHERE) {
        const double c3[3] = {a.x, a.y, a.z};
        for (int k = 0; k < 3; ++k) { lo[k] = c3[k] - a.w; hi[k] = c3[k] + a.w; }
        return;
    }
    const double v[3][3] = {{a.x, a.y, a.z}, {(double)a.x + b.x, (double)a.y + b.y, (double)a.z + b.z},
                            {(double)a.x + c.x, (double)a.y + c.y, (double)a.z + c.z}};
    for (int k = 0; k < 3; ++k) {
        lo[k] = std::min({v[0][k], v[1][k], v[2][k]});
        hi[k] = std::max({v[0][k], v[1][k], v[2][k]});
    }
}

static void check(const Scene &s, const ptbvh::Built &b, bool keyed, const char *what) {
    const int n_nodes = (int)b.nodes.size() / 2;
    std::vector<int> seen(s.types.size(), 0);
    for (int id : b.prim_of_slot)
        if (id >= 0) seen[id]++;
    for (int id : b.rest) seen[id] += 100;
    for (size_t i = 0; i < s.types.size(); ++i) {
        const bool in_tree_type = s.types[i] == b.type;
        CHECK(seen[i] == (in_tree_type ? 1 : 100), "%s: primitive %zu placed %d times", what, i, seen[i]);
    }
    CHECK(b.prim_of_slot.size() % ptbvh::kLeaf == 0 && b.prims.size() == 3 * b.prim_of_slot.size(), "%s: slot arrays", what);
    // walk the preorder layout: subtree of node i is [i, skip(i))
    int leaves = 0, partial = 0, last_key = -2, key_changes = 0;
    for (int i = 0; i < n_nodes; ++i) {
        const int skip = ptbvh::float_bits(b.nodes[2 * i].w), first = ptbvh::float_bits(b.nodes[2 * i + 1].w);
        CHECK(skip > i && skip <= n_nodes, "%s: node %d skip %d", what, i, skip);
        if (first < 0) {
            CHECK(i + 1 < n_nodes && skip > i + 1, "%s: inner node %d without children", what, i);
            const int left_skip = ptbvh::float_bits(b.nodes[2 * (i + 1)].w);
            CHECK(left_skip < skip && ptbvh::float_bits(b.nodes[2 * left_skip].w) == skip, "%s: node %d children do not tile its range", what, i);
            continue;
        }
        CHECK(skip == i + 1, "%s: leaf %d skip", what, i);
        CHECK(first == leaves * ptbvh::kLeaf, "%s: leaf %d first %d: leaves are not in slot order", what, i, first);
        ++leaves;
        int filled = 0, key = -2;
        for (int k = 0; k < ptbvh::kLeaf; ++k) {
            const int id = b.prim_of_slot[first + k];
            if (id < 0) continue;
            ++filled;
            CHECK(key == -2 || key == s.keys[id] || !keyed, "%s: leaf %d mixes keys", what, i);
            key = s.keys[id];
            double lo[3], hi[3];
            extent(s, id, lo, hi);
            // every ancestor (any node whose range covers this leaf) must contain the primitive
            for (int a = 0; a <= i; ++a) {
                if (ptbvh::float_bits(b.nodes[2 * a].w) <= i) continue;
                const float4 nlo = b.nodes[2 * a], nhi = b.nodes[2 * a + 1];
                CHECK(nlo.x <= lo[0] && nlo.y <= lo[1] && nlo.z <= lo[2] && nhi.x >= hi[0] && nhi.y >= hi[1] && nhi.z >= hi[2],
                      "%s: node %d does not contain primitive %d", what, a, id);
            }
        }
        CHECK(filled > 0, "%s: empty leaf %d", what, i);
        partial += filled < ptbvh::kLeaf;
        if (keyed && key != last_key) { ++key_changes; last_key = key; }
    }
    CHECK(leaves * ptbvh::kLeaf == (int)b.prim_of_slot.size(), "%s: %d leaves for %zu slots", what, leaves, b.prim_of_slot.size());
    if (keyed) {
        int distinct = 0;
        std::vector<int> have(64, 0);
        for (size_t i = 0; i < s.types.size(); ++i)
            if (s.types[i] == b.type && !have[s.keys[i] + 1]++) ++distinct;
        CHECK(key_changes == distinct, "%s: leaves are not laid out key by key (%d runs, %d keys)", what, key_changes, distinct);
        CHECK(partial <= 2 * distinct + 2, "%s: %d partly filled leaves for %d keys", what, partial, distinct);
    }
}

int main() {
    for (unsigned seed = 1; seed <= 6; ++seed) {
        const Scene a = random_scene(seed, 50 * seed + 3, 7, 0);
        check(a, ptbvh::build(a.prims, a.types), false, "triangles");
        const Scene b = random_scene(seed + 10, 5, 90 * seed, 0);
        check(b, ptbvh::build(b.prims, b.types), false, "spheres");
        const Scene c = random_scene(seed + 20, 200 * seed + 1, 6, 5);
        check(c, ptbvh::build(c.prims, c.types, &c.keys, PT_TRIANGLE), true, "keyed");
    }
    // degenerate input: identical centroids, a non-finite primitive, fewer primitives than a leaf
    Scene d;
    for (int i = 0; i < 9; ++i) {
        d.prims.push_back(make_float4(1, 1, 1, 0));
        d.prims.push_back(make_float4(1, 0, 0, 0));
        d.prims.push_back(make_float4(0, i == 4 ? NAN : 1, 0, 0));
        d.types.push_back(PT_TRIANGLE);
        d.keys.push_back(i % 2);
    }
    const ptbvh::Built plain = ptbvh::build(d.prims, d.types);
    CHECK(plain.rest.size() == 1 && plain.rest[0] == 4, "non-finite primitive must go to the rest list");
    const ptbvh::Built keyed = ptbvh::build(d.prims, d.types, &d.keys, PT_TRIANGLE);
    CHECK(keyed.rest.empty(), "keyed build keeps every primitive of its type in the tree");
    Scene e = random_scene(99, 2, 0, 0);
    check(e, ptbvh::build(e.prims, e.types), false, "tiny");
    std::printf(failures ? "%d failures\n" : "ok\n", failures);
    return failures != 0;
}
