// Per-lane closest-hit search through the LBVH, shared by the ray-buffer kernel (traverse.hip) and
// the path-tracing kernel (path_trace.hip). Result-equivalent to rayCastLBVH<false>
// (src/ray_tracer/ray_tracer_render.cu:192-270):
//   * the same root test, the same slab test per internal child (include/cg/ray.h:29-45) and the same
//     two-sided Moeller-Trumbore per triangle (3rd/glm/gtx/intersect.inl:28-94), with a * b + c never
//     fused;
//   * leaf children carry no box in the reference's node layout, so their triangles are always tested;
//   * the reference walks left-then-right without distance culling and keeps the FIRST hit among equal
//     t (strict <). Its walk meets triangles in increasing position k in `faces`, so "first" is
//     "smallest k". Here the nearer child is entered first and boxes entered beyond the best hit are
//     skipped; equal t is resolved towards the smaller k, which gives the reference's winner.
#pragma once
#include "kk_math.h"

#include "../../include/kouek_hip.h"

namespace kk {
namespace trav {

constexpr uint32_t LEAF = 0x80000000u;
constexpr uint32_t LEAF_MASK = 0x7fffffffu;
constexpr int STACK_DEPTH = 32; // internal-node depth <= 30: every level fixes one more Morton bit
// A box is skipped when its entry distance exceeds the best hit by more than this relative slack;
// the slack covers the few-ulp disagreement between slab and triangle arithmetic on a box face.
constexpr float CULL_SLACK = 1e-5f;

struct Scene {
    const kk_lbvh_node *nodes;
    const uint32_t *faces;
    const uint32_t *compacted;
    const float *positions;
    const uint32_t *tri;
    uint32_t faceNum, compactedNum;
};

struct Hit {
    float t;
    float bx, by;
    uint32_t fi; // face index, 0xFFFFFFFF = miss
    uint32_t k;  // position in `faces`
};

struct NodeData {
    uint32_t lft, rht;
    AABB box;
};

KK_D NodeData load_node(const kk_lbvh_node *__restrict__ nodes, uint32_t i) {
    const uint4 *p = reinterpret_cast<const uint4 *>(nodes + i);
    const uint4 a = p[0], b = p[1];
    NodeData n;
    n.lft = a.x;
    n.rht = a.y;
    n.box.minPos = {__uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x)};
    n.box.maxPos = {__uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)};
    return n;
}

KK_D f3 load_pos(const float *__restrict__ p, uint32_t i) {
    return {p[3ull * i], p[3ull * i + 1], p[3ull * i + 2]};
}

template <bool COUNT>
KK_D void test_leaf(const Scene &s, const Ray &ray, uint32_t leaf, Hit &best, uint32_t &triTests) {
    const uint32_t c = leaf & LEAF_MASK;
    uint32_t k = s.compacted[c];
    const uint32_t kEnd = (c == s.compactedNum - 1) ? s.faceNum : s.compacted[c + 1];
    for (; k < kEnd; ++k) {
        const uint32_t fi = s.faces[k];
        const uint32_t i0 = s.tri[3ull * fi], i1 = s.tri[3ull * fi + 1], i2 = s.tri[3ull * fi + 2];
        float bx = 0.f, by = 0.f;
        const float t = HitTriangle(ray, load_pos(s.positions, i0), load_pos(s.positions, i1),
                                    load_pos(s.positions, i2), bx, by);
        if (COUNT)
            ++triTests;
        if (t >= 0.f && (t < best.t || (t == best.t && k < best.k))) {
            best.t = t;
            best.bx = bx;
            best.by = by;
            best.fi = fi;
            best.k = k;
        }
    }
}

// Walk state of one lane. `stack` points at this lane's first LDS slot; consecutive entries are
// `stride` words apart so that a wave's accesses fall on consecutive banks.
struct Walk {
    uint32_t lft, rht; // children of the node being visited
    int sp;
    bool done;
};

template <bool COUNT>
KK_D void walk_begin(const Scene &s, const Ray &ray, Walk &w, Hit &best, uint32_t &nodeFetches) {
    best.t = FloatMax;
    best.bx = best.by = 0.f;
    best.fi = 0xFFFFFFFFu;
    best.k = 0xFFFFFFFFu;
    const NodeData root = load_node(s.nodes, 0);
    if (COUNT)
        ++nodeFetches;
    float tEnter, tExit;
    HitAABB(ray, root.box, tEnter, tExit);
    w.sp = 0;
    w.lft = root.lft;
    w.rht = root.rht;
    w.done = tEnter > tExit || !ray_is_finite(ray.pos, ray.dir);
}

// Visits one internal node: tests its leaf children's triangles, box-tests its internal children,
// then moves to the nearer surviving child (pushing the other) or pops.
template <bool COUNT>
KK_D void walk_step(const Scene &s, const Ray &ray, Walk &w, Hit &best, uint32_t *stack, int stride,
                    uint32_t &nodeFetches, uint32_t &triTests) {
    const uint32_t lft = w.lft, rht = w.rht;
    if (lft & LEAF)
        test_leaf<COUNT>(s, ray, lft, best, triTests);
    if ((rht & LEAF) && rht != lft)
        test_leaf<COUNT>(s, ray, rht, best, triTests);

    bool hitL = false, hitR = false;
    float enterL = 0.f, enterR = 0.f;
    NodeData nl, nr;
    if (!(lft & LEAF)) {
        nl = load_node(s.nodes, lft);
        if (COUNT)
            ++nodeFetches;
        float tExit;
        HitAABB(ray, nl.box, enterL, tExit);
        hitL = !(enterL > tExit);
    }
    if (!(rht & LEAF)) {
        nr = load_node(s.nodes, rht);
        if (COUNT)
            ++nodeFetches;
        float tExit;
        HitAABB(ray, nr.box, enterR, tExit);
        hitR = !(enterR > tExit);
    }
    const float limit = best.t + CULL_SLACK * best.t;
    hitL = hitL && !(enterL > limit);
    hitR = hitR && !(enterR > limit);

    if (hitL && hitR && w.sp < STACK_DEPTH) {
        const bool leftFirst = !(enterR < enterL);
        stack[w.sp * stride] = leftFirst ? rht : lft;
        ++w.sp;
        w.lft = leftFirst ? nl.lft : nr.lft;
        w.rht = leftFirst ? nl.rht : nr.rht;
    } else if (hitL) {
        w.lft = nl.lft;
        w.rht = nl.rht;
    } else if (hitR) {
        w.lft = nr.lft;
        w.rht = nr.rht;
    } else if (w.sp > 0) {
        --w.sp;
        const uint32_t n = stack[w.sp * stride];
        const uint2 ch = *reinterpret_cast<const uint2 *>(s.nodes + n);
        w.lft = ch.x;
        w.rht = ch.y;
    } else
        w.done = true;
}

// Whole search for one ray (used where a lane traces rays one after another).
template <bool COUNT>
KK_D Hit closest_hit(const Scene &s, const Ray &ray, uint32_t *stack, int stride,
                     uint32_t &nodeFetches, uint32_t &triTests) {
    Walk w;
    Hit best;
    walk_begin<COUNT>(s, ray, w, best, nodeFetches);
    while (!w.done)
        walk_step<COUNT>(s, ray, w, best, stack, stride, nodeFetches, triTests);
    return best;
}

} // namespace trav
} // namespace kk
