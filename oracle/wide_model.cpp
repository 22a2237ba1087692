// ORACLE -- test infrastructure only.  A CPU model of the device's 4-WIDE traversal order (csrc/device_trace.h,
// node phase of the timed kernels), used by tests/test_wide_model.py to check the exactness argument of
// DESIGN.md section 4 against the reference traversal (bvh.rs:243-300) on the CPU, ray by ray:
//
//   the 4-wide walk tests the SAME triangles, in the SAME order, with the SAME running t_max
//
// as traverse_bvh(Nearest), so hit, primitive and t are the reference's bits also where two hits tie or where a
// box test and a triangle test disagree by a rounding.  Both walks are restated here over the flattened arrays
// (rt_scene_flat): the binary one follows bvh.rs line by line, the wide one follows the device code.
//
// A wide node W(P) of an internal node P holds four slots: the children of P's first child (slots 0, 1) and of
// its second child (slots 2, 3); a child of P that is a leaf sits in the even slot of its group and the odd slot
// is empty (lower = +inf, upper = -inf: never hit).  The intermediate boxes -- the two children of P -- are not
// stored: a child's box is the union of its children's boxes bound for bound (the builder grows every node's box
// over its primitives with exact min/max, bvh.rs:89-100), so its near planes are rebuilt from theirs.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace {

struct Flat {
    const float *upper, *lower;       // [n_nodes][3]
    const uint32_t *index, *num_prim; // [n_nodes]
    const float *v0, *v1, *v2;        // [n_tris][3], leaf order
};

// AxisAlignedBoundingBox::hit (primitives.rs:304-321); also returns the per-axis near distances
inline bool aabb(const float* lo, const float* hi, const float* o, const float* inv, float t_min, float t_max, float* entry,
                 float* nears) {
    for (int a = 0; a < 3; ++a) {
        float t0 = (lo[a] - o[a]) * inv[a], t1 = (hi[a] - o[a]) * inv[a];
        if (inv[a] < 0.0f) std::swap(t0, t1);
        nears[a] = t0;
        t_min = fmaxf(t_min, t0);  // f32::max / f32::min ignore NaN like fmaxf / fminf
        t_max = fminf(t_max, t1);
    }
    *entry = t_min;
    return !(t_max <= t_min);
}

// triangle_intersection (primitives.rs:138-186)
inline bool tri(const Flat& f, uint32_t k, const float* o, const float* d, float t_min, float t_max, float* t_out) {
    const float *a = f.v0 + 3 * (size_t)k, *b = f.v1 + 3 * (size_t)k, *c = f.v2 + 3 * (size_t)k;
    float e1[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, e2[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    float p[3] = {d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0]};
    float det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (fabsf(det) < 1.1920929e-7f) return false;
    float inv_det = 1.0f / det;
    float s[3] = {o[0] - a[0], o[1] - a[1], o[2] - a[2]};
    float u = inv_det * (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]);
    if (u < 0.0f || u > 1.0f) return false;
    float q[3] = {s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0]};
    float v = inv_det * (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]);
    if (v < 0.0f || u + v > 1.0f) return false;
    float t = inv_det * (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]);
    if (t > t_min && t < t_max) {
        *t_out = t;
        return true;
    }
    return false;
}

struct Trace {  // what a walk did: an order-sensitive hash of (triangle tested, t_max it was tested with)
    uint64_t hash = 1469598103934665603ull;
    uint32_t n_tri = 0, n_fetch = 0, hit = 0xFFFFFFFFu, max_stack = 0;
    float t = INFINITY;
    void test(uint32_t k, float t_max) {
        uint32_t b;
        memcpy(&b, &t_max, 4);
        hash = (hash ^ k) * 1099511628211ull;
        hash = (hash ^ b) * 1099511628211ull;
        n_tri += 1;
    }
};

void leaf(const Flat& f, uint32_t node, const float* o, const float* d, float t_min, float* t_max, Trace* tr) {
    for (uint32_t i = f.index[node]; i < f.index[node] + f.num_prim[node]; ++i) {
        tr->test(i, *t_max);
        float t;
        if (tri(f, i, o, d, t_min, *t_max, &t)) {
            *t_max = t;
            tr->hit = i;
            tr->t = t;
        }
    }
}

// bvh.rs:243-300, Nearest
void walk_binary(const Flat& f, const float* o, const float* d, float t_min, float t_max, Trace* tr) {
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    std::vector<uint32_t> st{0};
    while (!st.empty()) {
        uint32_t n = st.back();
        st.pop_back();
        if (f.num_prim[n] > 0) {
            leaf(f, n, o, d, t_min, &t_max, tr);
            continue;
        }
        tr->n_fetch += 1;
        uint32_t c1 = f.index[n], c2 = c1 + 1;
        float t1, t2, nn[3];
        bool h1 = aabb(f.lower + 3 * (size_t)c1, f.upper + 3 * (size_t)c1, o, inv, t_min, t_max, &t1, nn);
        bool h2 = aabb(f.lower + 3 * (size_t)c2, f.upper + 3 * (size_t)c2, o, inv, t_min, t_max, &t2, nn);
        if (!h1 && !h2) continue;
        if (h1 && !h2) { st.push_back(c1); continue; }
        if (!h1 && h2) { st.push_back(c2); continue; }
        if (t1 > t2) std::swap(c1, c2);
        st.push_back(c2);
        st.push_back(c1);
        if (st.size() > tr->max_stack) tr->max_stack = (uint32_t)st.size();
    }
}

// The device's wide walk.  Stack words: a plain entry is one word (node), a DEFERRED entry two (entry distance, node):
// the children of P's far child are box-tested by the reference when that child is popped, with the t_max of that
// moment; the wide walk tested them early, with a t_max at least as large, and repeats the test at the pop.  The test
// !(min(T', exit) <= entry) is !(T' <= entry) && !(exit0 <= entry) with exit0 free of t_max, and the second factor held
// when the entry was pushed: the entry distance alone decides.
constexpr uint32_t DEFER = 0x40000000u, PAIR = 0x20000000u;
void walk_wide(const Flat& f, const float* o, const float* d, float t_min, float t_max, Trace* tr, int defer_internal) {
    float inv[3] = {1.0f / d[0], 1.0f / d[1], 1.0f / d[2]};
    std::vector<uint32_t> st;
    uint32_t cur = 0;
    bool have = true;
    auto pop_next = [&]() {
        have = false;
        while (!st.empty()) {
            uint32_t v = st.back();
            st.pop_back();
            if (v & DEFER) {
                // the moment the reference pops the far child of P and box-tests ITS children, both with the t_max
                // of this moment: the near one is visited now, the far one goes back as a plain entry (a popped
                // node is not tested again, bvh.rs:251-253)
                float e;
                memcpy(&e, &st.back(), 4);
                st.pop_back();
                const bool near_ok = !(t_max <= e);
                uint32_t far = 0;
                bool far_ok = false;
                if (v & PAIR) {
                    far = st.back();
                    st.pop_back();
                    float ef;
                    memcpy(&ef, &st.back(), 4);
                    st.pop_back();
                    far_ok = !(t_max <= ef);
                }
                v &= ~(DEFER | PAIR);
                if (near_ok && far_ok) st.push_back(far);
                else if (far_ok) v = far;
                else if (!near_ok) continue;
            }
            cur = v;
            have = true;
            return;
        }
    };
    while (have) {
        if (f.num_prim[cur] > 0) {
            leaf(f, cur, o, d, t_min, &t_max, tr);
            pop_next();
            continue;
        }
        tr->n_fetch += 1;
        // slots
        uint32_t slot[4];
        bool valid[4], direct[2];
        for (int g = 0; g < 2; ++g) {
            uint32_t c = f.index[cur] + g;
            direct[g] = f.num_prim[c] > 0;
            if (direct[g]) {
                slot[2 * g] = c, valid[2 * g] = true, valid[2 * g + 1] = false, slot[2 * g + 1] = 0;
            } else {
                slot[2 * g] = f.index[c], slot[2 * g + 1] = f.index[c] + 1, valid[2 * g] = valid[2 * g + 1] = true;
            }
        }
        bool h[4];
        float e[4], nn[3];
        const float pinf[3] = {INFINITY, INFINITY, INFINITY}, ninf[3] = {-INFINITY, -INFINITY, -INFINITY};
        const float *lo[4], *hi[4];
        for (int s = 0; s < 4; ++s) {
            lo[s] = valid[s] ? f.lower + 3 * (size_t)slot[s] : pinf;
            hi[s] = valid[s] ? f.upper + 3 * (size_t)slot[s] : ninf;
            h[s] = aabb(lo[s], hi[s], o, inv, t_min, t_max, &e[s], nn) && valid[s];
        }
        const bool g1 = h[0] || h[1], g2 = h[2] || h[3];
        // entry distance of the two intermediate boxes (the children of P): the near planes of the union of a group's
        // slots through the reference's slab arithmetic, as the device computes it
        float E[2];
        for (int g = 0; g < 2; ++g) {
            float en = t_min;
            for (int a = 0; a < 3; ++a) {
                const float plane = inv[a] < 0.0f ? fmaxf(hi[2 * g][a], hi[2 * g + 1][a]) : fminf(lo[2 * g][a], lo[2 * g + 1][a]);
                en = fmaxf(en, (plane - o[a]) * inv[a]);
            }
            E[g] = en;
        }
        const int A = (g2 && (!g1 || E[0] > E[1])) ? 1 : 0, B = A ^ 1;  // group visited first / second
        struct Ent { uint32_t node; float e; bool ok; };
        auto order = [&](int g, Ent* nearo, Ent* faro) {
            const int s0 = 2 * g, s1 = 2 * g + 1;
            const bool both = h[s0] && h[s1], sw = both && e[s0] > e[s1];
            const int ns = (h[s1] && (!h[s0] || sw)) ? s1 : s0, fs = sw ? s0 : s1;
            *nearo = {slot[ns], e[ns], h[s0] || h[s1]};
            *faro = {slot[fs], e[fs], both};
        };
        Ent nA, fA, nB, fB;
        order(A, &nA, &fA);
        order(B, &nB, &fB);
        const bool deferB = !direct[B] && nA.ok;
        auto word = [](float x) {
            uint32_t b;
            memcpy(&b, &x, 4);
            return b;
        };
        auto is_leaf = [&](const Ent& x) { return f.num_prim[x.node] > 0; };
        if (nB.ok && nA.ok) {
            // group B's slots are the children of P's FAR child: the reference tests them when it pops that child.
            // A leaf among them has to be tested again then (its triangles may lie behind a hit found meanwhile and the
            // reference would not test them); an internal one need not (its own slots all fail, see the root-box skip).
            const bool pair = deferB && fB.ok && (is_leaf(fB) || defer_internal);
            if (pair) {
                st.push_back(word(fB.e));
                st.push_back(fB.node);
                st.push_back(word(nB.e));
                st.push_back(nB.node | DEFER | PAIR);
            } else {
                if (fB.ok) st.push_back(fB.node);
                if (deferB && (is_leaf(nB) || defer_internal)) {
                    st.push_back(word(nB.e));
                    st.push_back(nB.node | DEFER);
                } else {
                    st.push_back(nB.node);
                }
            }
        }
        if (fA.ok) st.push_back(fA.node);
        if (st.size() > tr->max_stack) tr->max_stack = (uint32_t)st.size();
        if (nA.ok) cur = nA.node;
        else pop_next();
    }
}

}  // namespace

// out[i] = {same (1/0), binary fetches, wide fetches, triangle tests binary, triangle tests wide, hit binary, hit wide,
//           max stack binary, max stack words wide}
extern "C" int wide_model_compare(const float* upper, const float* lower, const uint32_t* index, const uint32_t* num_prim,
                                  const float* v0, const float* v1, const float* v2, const float* org, const float* dir,
                                  size_t n, float t_min, float t_max, int defer_internal, uint32_t* out) {
    Flat f{upper, lower, index, num_prim, v0, v1, v2};
    int bad = 0;
    for (size_t i = 0; i < n; ++i) {
        Trace a, b;
        walk_binary(f, org + 3 * i, dir + 3 * i, t_min, t_max, &a);
        walk_wide(f, org + 3 * i, dir + 3 * i, t_min, t_max, &b, defer_internal);
        const bool same = a.hash == b.hash && a.n_tri == b.n_tri && a.hit == b.hit && memcmp(&a.t, &b.t, 4) == 0;
        uint32_t* r = out + 9 * i;
        r[0] = same, r[1] = a.n_fetch, r[2] = b.n_fetch, r[3] = a.n_tri, r[4] = b.n_tri, r[5] = a.hit, r[6] = b.hit;
        r[7] = a.max_stack, r[8] = b.max_stack;
        bad += !same;
    }
    return bad;
}
